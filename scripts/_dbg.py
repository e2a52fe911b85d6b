import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ILQG_DEBUG_ROWPROG"] = "1"
from ilqgames_amd import abi, examples, hip
for name in examples.CONFIGS:
    spec = examples.CONFIGS[name]()
    print(name, flush=True)
    try:
        hip.Problem(spec, abi.F64)
    except Exception as e:
        print("  ", e)
