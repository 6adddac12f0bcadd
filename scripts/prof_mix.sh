cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_mix
rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/prof_mix -o mix -- python scripts/mix_probe.py > gpurun_out/prof_mix/out.txt 2>&1
cat gpurun_out/prof_mix/out.txt | tail -3
f=$(find gpurun_out/prof_mix -name "*kernel_stats.csv" | head -1); cat $f | cut -c1-150
