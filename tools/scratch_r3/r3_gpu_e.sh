#!/bin/bash
mkdir -p gpurun_out/r3
hipcc -O3 --offload-arch=gfx950 -Wno-unused-result tools/ubench_xstream.hip -o /tmp/ubench_xstream 2>/dev/null && timeout -k 5 100 /tmp/ubench_xstream > gpurun_out/r3/xstream.txt 2>&1; cat gpurun_out/r3/xstream.txt
