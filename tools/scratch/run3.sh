python -m pytest tests/test_gpu_estep.py -x -q -m gpu -k "streamed_slots" 2>&1 | tail -5
for o in 1 2; do
python tools/class_ab.py cfg4 225 240 quad_stream=$o 2>&1 | tail -1
python tools/class_ab.py cfg4 225 232 quad_stream=$o 2>&1 | tail -1
python tools/class_ab.py cfg4 241 256 quad_stream=$o 2>&1 | tail -1
python tools/class_ab.py cfg3 225 256 quad_stream=$o 2>&1 | tail -1
done
