import os, sys
sys.path.insert(0, os.getcwd())
import sys, runpy
sys.argv = ['bench_train.py', '--steps', '30', '--graph', '--parts', 'd']
from go_with_the_flows_amd import _lib
import os
if os.environ.get('SMALL') == '1':
    _lib.set_tuning(_lib.TUNE_SMALL_LIGHT_TILE)
runpy.run_path('tools/bench_train.py', run_name='__main__')
