ulimit -c 0; mkdir -p gpurun_out/r04
( for op in findall extract; do python tools/probe_op.py $op 2>&1 | tail -1; done; python -m pytest tests -m gpu -x -q -k "extract or findall or chain or golden or parity" 2>&1 | grep -E "^E|passed|failed|Error" | tail -5 ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/q1.txt
