#!/bin/bash
# Long parity runs of the final build on the GPU box (run through gpurun): GPU and oracle side by side, 2-/3-rank sharded
# driver against the single engine, and the north-star run shape.  Summaries under gpurun_out/, copied to profiles/<round>_*.
set -u
mkdir -p gpurun_out
timeout -k 10 200 python tests/soak.py cube_1e5 3000 500 > gpurun_out/soak_c5.log 2>&1; tail -1 gpurun_out/soak_c5.log | cut -c1-330
timeout -k 10 300 python tests/soak.py cube_1e6 200 100 > gpurun_out/soak_c6.log 2>&1; tail -1 gpurun_out/soak_c6.log | cut -c1-330
timeout -k 10 300 python tests/soak.py temp_1e6 100 50 > gpurun_out/soak_t6.log 2>&1; tail -1 gpurun_out/soak_t6.log | cut -c1-380
timeout -k 10 200 python tests/soak_sharded.py cube_1e6 100 2 > gpurun_out/soak_s2c.log 2>&1; tail -1 gpurun_out/soak_s2c.log | cut -c1-330
timeout -k 10 200 python tests/soak_sharded.py pore_1e6 200 2 > gpurun_out/soak_s2p.log 2>&1; tail -1 gpurun_out/soak_s2p.log | cut -c1-330
timeout -k 10 200 python tests/soak_sharded.py cube_1e5 500 3 > gpurun_out/soak_s3c.log 2>&1; tail -1 gpurun_out/soak_s3c.log | cut -c1-330
timeout -k 10 200 python tests/long_run.py pore_1e6 10000 > gpurun_out/long.log 2>&1; tail -1 gpurun_out/long.log | cut -c1-400
