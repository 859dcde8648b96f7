cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_knn.py "tests/test_gpu_frame_stream.py::test_pipelined_frames_enqueued_back_to_back" -x -q 2>&1 | tail -5
B="python bench.py --words 125000 --steps 200 --warmup 20 --no-cpu-baseline --no-extras"
for px in 0 248 163 0 248 256; do echo "px=$px"; LCD_BF_PX=$px timeout 300 $B 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['roofline'].get('kernel'), d['roofline'].get('launch_us', d['roofline']))
"; done
for px in 0 248; do echo "1M px=$px"; LCD_BF_PX=$px timeout 300 python bench.py --words 1000000 --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['roofline'])
"; done
