python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r02_t4.log; tail -15 gpurun_out/r02_t4.log
for m in 0 1 2 4; do LM_DEBUG=1 LM_SP_DEBUG_MODE=$m python -c "
import lexicmap_amd as la
la.Index.synthetic(20000, 2000000, 1001).close()" 2>&1 | grep "partition sort\|builder pass 1"; done
LM_DEBUG=1 timeout 600 python bench.py --workload c3mini --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02_b5.json 2> gpurun_out/r02_b5.err; grep -i "error\|Traceback\|wfa pass" gpurun_out/r02_b5.err | sort | uniq -c | sort -rn | head; tail -c 300 gpurun_out/r02_b5.json
timeout 600 python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02_b6.json 2> gpurun_out/r02_b6.err; grep -i "error\|Traceback" gpurun_out/r02_b6.err | head; tail -c 300 gpurun_out/r02_b6.json
