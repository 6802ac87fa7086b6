import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["OFK_STREAMS"]="1"; os.environ["OFK_NO_OVERLAP"]="1"
sys.argv=["x"]
import importlib.util
spec=importlib.util.spec_from_file_location("bc", os.path.join(sys.path[0],"tools/bench_configs.py")); bc=importlib.util.module_from_spec(spec); spec.loader.exec_module(bc)
from __graft_entry__ import load_package; load_package()
from of_amd.pipeline import PipelineConfig
for mc, md in ((2000,10),(500,10),(2000,0),(1000,10),(4000,5)):
    bc.run(f"C5 mc={mc} md={md}", 3840, 2160, 32, PipelineConfig(max_corners=mc, quality=0.01, min_distance=md, block_size=7, win=15, max_level=5), 6)
