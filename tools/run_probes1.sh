#!/bin/bash
mkdir -p gpurun_out/r04
python tools/probes/strips_soak.py 400 > gpurun_out/r04/r_soak.txt 2>&1
python tools/probes/strips_soak.py 120 nopeers > gpurun_out/r04/r_soak_nopeers.txt 2>&1
python tools/probes/host_cost_probe.py 256 144 > gpurun_out/r04/r_host_cost.txt 2>&1
python tools/probes/host_cost_probe.py 1920 171 >> gpurun_out/r04/r_host_cost.txt 2>&1
python tools/probes/deform_soak.py bunny.obj 1920 1080 600 0.3 100.0 16 > gpurun_out/r04/r_deform.txt 2>&1
python tools/probes/deform_soak.py bunny.obj 1920 1080 600 0.3 1.02 16 >> gpurun_out/r04/r_deform.txt 2>&1
python tools/probes/deform_soak.py dragon.obj 1920 1080 600 0.3 100.0 16 >> gpurun_out/r04/r_deform.txt 2>&1
python tools/probes/deform_soak.py dragon.obj 1920 1080 600 0.3 1.02 16 >> gpurun_out/r04/r_deform.txt 2>&1
tail -3 gpurun_out/r04/r_soak.txt; cat gpurun_out/r04/r_host_cost.txt; grep "deform" gpurun_out/r04/r_deform.txt
