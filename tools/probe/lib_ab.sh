# two (or more) builds of the library on the encoder timeline, alternating on one box.  Usage: bash tools/probe/lib_ab.sh variants/lib_a.so variants/lib_b.so [pattern]
cd /root/repo
PAT=${3:-up_kernel|kernel time}
for R in 1 2 3; do
  for L in "$1" "$2"; do
    echo "== $L"
    if [ "$L" = "product" ]; then bash tools/probe/enc_tl.sh ab | grep -E "$PAT"; else VTACO_HIP_LIB=$L bash tools/probe/enc_tl.sh ab | grep -E "$PAT"; fi
  done
done
