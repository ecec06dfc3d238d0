import json, sys
a=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); b=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], a["value"], sys.argv[2], b["value"])
kb={k["cfg"]:k for k in b["kernels"]}
for k in a["kernels"]:
    print(k["cfg"], k["avg_launch_us"], kb[k["cfg"]]["avg_launch_us"], k["kernel"][:50])
