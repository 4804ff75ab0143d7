set -x
python -m pytest tests/test_gpu_estep.py -x -q -m gpu -k "streamed_slots or wide_table or every_kernel" 2>&1 | tail -15
for o in 0 1; do python tools/class_ab.py cfg4 225 256 quad_stream=$o 2>&1 | tail -1; done
for o in 0 1; do python tools/class_ab.py cfg4 225 240 quad_stream=$o 2>&1 | tail -1; done
python tools/class_ab.py cfg4 209 224 2>&1 | tail -1
for o in 0 1; do python tools/class_ab.py cfg3 225 256 quad_stream=$o 2>&1 | tail -1; done
