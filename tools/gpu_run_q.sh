#!/bin/bash
mkdir -p gpurun_out/r2q
timeout 900 python tools/q30_probe.py > gpurun_out/r2q/q30_probe.log 2>&1
tail -40 gpurun_out/r2q/q30_probe.log
