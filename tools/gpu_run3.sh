#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch; p=torch.cuda.get_device_properties(0); print(p.name, p.multi_processor_count, p.total_memory)" > gpurun_out/dev.log 2>&1
for i in 1 2; do
(timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -12) > gpurun_out/pytest_gpu_$i.log 2>&1
done
(timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline) > gpurun_out/bench_fused.log 2>&1
cat gpurun_out/dev.log; tail -12 gpurun_out/pytest_gpu_1.log; tail -3 gpurun_out/pytest_gpu_2.log; tail -3 gpurun_out/bench_fused.log
