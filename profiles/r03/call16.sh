#!/bin/bash
# call 16 (last of round 3): the host-side changes made after call 15 (Decompresser skip rule, header checks for blocks
# without a model, Compressor guards) on the GPU paths of the C++ API and the reference's own archiver
mkdir -p gpurun_out/r03c16
timeout 200 bash -c 'python -c "import __graft_entry__ as g; g.smoke()" && python -m pytest tests/test_cpp_api.py tests/test_cli.py -m gpu -x -q --durations=8' > gpurun_out/r03c16/out.txt 2>&1
echo "rc=$?" >> gpurun_out/r03c16/out.txt
tail -25 gpurun_out/r03c16/out.txt
