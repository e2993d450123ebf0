ulimit -c 0; mkdir -p gpurun_out/r04
( python tools/bench_ops.py --only C3 2>/dev/null | grep '^{' | head -4 | cut -c1-100; python -m pytest tests -m gpu -x -q -k "regex or chain or extract or findall or count or contains or match" 2>&1 | grep -E "^E|passed|failed|Error" | tail -5 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/q1.txt
