#!/bin/bash
# one gpurun call: s_memtime stations of one wave of k_filter_stream2, uniform reads against views (RV form)
out=gpurun_out/r05views; mkdir -p $out
export CAH_LIB_PATH=$PWD/cutadapt_amd/libcutadapt_hip_trace.so
PYTHONPATH=$PWD python profiles/scripts/s2_trace.py 100000000 > $out/trace_uniform.txt 2>&1
PYTHONPATH=$PWD python profiles/scripts/s2_trace.py 100000000 views > $out/trace_views.txt 2>&1
tail -3 $out/trace_uniform.txt; tail -3 $out/trace_views.txt
