#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2h
mkdir -p $O
./tools/coexec_probe > $O/coexec.log 2>&1
python tools/emit_probe.py > $O/emit_probe.log 2>&1
cat $O/coexec.log $O/emit_probe.log
