# final validation on the GPU box: parity tests, smoke, the 2-rank spawn path (gloo, one GPU)
python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r04_tests_gpu.log; grep -v "RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl" gpurun_out/r04_tests_gpu.log | tail -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python bench.py --gpus 2 --dist-backend gloo --workload tiny --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_g2.json 2> gpurun_out/r04_g2.err; grep -i "error\|Traceback" gpurun_out/r04_g2.err | head -5; python -c "
import json;d=json.loads(open('gpurun_out/r04_g2.json').read().strip().splitlines()[-1]);print('n_gpus',d['n_gpus'],d['value'],d['unit'],d['scaling'],d['rows'])"
