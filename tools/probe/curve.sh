for b in 5 6; do echo "hilbert bits $b"; BTR_FPS_GRID_BITS=$b python tools/fps_ab.py default 2>&1 | tail -3; BTR_FPS_GRID_BITS=$b python tools/fps_prof.py 2>&1 | grep -E "touched buckets/step total"; done
python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "fps or furthest" 2>&1 | tail -2
BTR_FPS_GRID_BITS=6 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "fps or furthest" 2>&1 | tail -2
