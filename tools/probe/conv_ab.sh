# A/B of the split-f16 conv variants (hw = specialised waves, hx = support work in the tap waves' gaps) on one box
cd /root/repo
for S in 1 2 1 2; do VTACO_CONV_SPEC=$S timeout 300 python3 tools/probe/conv_ab.py; done > gpurun_out/conv_ab.txt 2>&1
AB_B=2 VTACO_CONV_SPEC=1 timeout 300 python3 tools/probe/conv_ab.py >> gpurun_out/conv_ab.txt 2>&1
AB_B=2 VTACO_CONV_SPEC=2 timeout 300 python3 tools/probe/conv_ab.py >> gpurun_out/conv_ab.txt 2>&1
cat gpurun_out/conv_ab.txt
