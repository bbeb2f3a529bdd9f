import cProfile, pstats, sys, os, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['loader_bench.py', '10000']
cProfile.run("runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'loader_bench.py'), run_name='__main__')", '/tmp/lp.prof')
pstats.Stats('/tmp/lp.prof').sort_stats('cumulative').sort_stats('tottime').print_stats(16)
