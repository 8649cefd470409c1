# A/B on one box: lattice tiles claimed from the workgroup's counter (default) against the fixed share per wave (VTACO_DECODE_CLAIM=0)
cd /root/repo
for i in 1 2; do
for c in 1 0; do
  for p in f16x3 f16f8 bf16x3 f32; do
    echo -n "claim=$c $p: "; VTACO_DECODE_CLAIM=$c python tools/diag_wg.py $p 2>&1 | grep "per launch" | sed 's/; last launch.*span_us/ span_us/' | cut -c1-120
  done
done
done
