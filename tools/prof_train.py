import cProfile, pstats, sys, os, runpy, io
sys.argv = ["train_generator.py"] + sys.argv[1:]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(os.environ["GRAFT_REPO_ROOT"], "train_generator.py"), run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
