#!/bin/bash
O=gpurun_out/r2ab; mkdir -p $O; rm -f $O/ab.txt
for v in "$@"; do
  if [ $v = cur ]; then unset FXPLAN_SO; else export FXPLAN_SO=$PWD/tools/probe_build/libfxplan_$v.so; fi
  echo "== $v" >> $O/ab.txt
  timeout 300 python3 tools/c3.py c3B c3A m1o >> $O/ab.txt 2>&1
done
cat $O/ab.txt
