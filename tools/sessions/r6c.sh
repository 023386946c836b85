set -u
O=gpurun_out/r6c; mkdir -p $O
W16=1 timeout 1200 bash tools/pmc_h2.sh "m.P4.bneck,m.P3.bneck,pose.P3.bneck,m.P5.bneck" > $O/conv_h2r_pmc.txt 2>&1
grep -v '^pass' $O/conv_h2r_pmc.txt | head -40
