#!/bin/bash
# round 5, call 1: operand-feed microbenchmarks (quad-k b128 layout vs the round-4 form) + the new full-size FreqCodec parity tests
mkdir -p gpurun_out/r5
cd tests/micro
./bin/conv_loop_feed > ../../gpurun_out/r5/micro_feed_r4form.txt 2>&1
./bin/conv_loop_feed_b128 > ../../gpurun_out/r5/micro_feed_b128.txt 2>&1
cd ../..
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "freqmpgr1 or gr1_benchmark or mag_angle or native_library" > gpurun_out/r5/pytest_new.log 2>&1
tail -5 gpurun_out/r5/pytest_new.log
cat gpurun_out/r5/micro_feed_b128.txt
