"""cProfile of the Python around the one-call path (bench's timed loop): where the host time outside ovm_infer goes."""
import cProfile, pstats, sys, os, io
sys.argv = ["bench.py", "--no-alt", "--no-cpu-baseline", "--steps", "60", "--warmup", "8"]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
finally:
    pr.disable()
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s).sort_stats("cumulative")
    st.print_callees("rcnn3d.py:.*(inference|_infer_fused|preprocess_image|_postprocess)")
    st.print_callees("_instances_from_records|records_to_fields|make_images|infer_gdino")
    print(s.getvalue()[:14000], file=sys.stderr)
