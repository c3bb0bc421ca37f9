import os, sys, runpy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from c3_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
