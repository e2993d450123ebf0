cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tok
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tok -o tok -- python tools/probe_tok.py > gpurun_out/tok/run.log 2>&1
python tools/kstats.py $(ls gpurun_out/tok/*/tok_kernel_stats.csv gpurun_out/tok/tok_kernel_stats.csv 2>/dev/null | head -1) 12
cat gpurun_out/tok/run.log | tail -5
