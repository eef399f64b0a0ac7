import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().split("\n")[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    a=d["all_steps"]
    print(f.split("/")[-1], "step_ms %.3f world_step %.3f host_structure %.3f solve_device %.3f launches %.0f | rebuilt %d placed %d created %d"%(a["step_ms"],a["world_step_ms"],a["host_structure_ms"],a["solve_device_ms"],a["launches"],d["steps_that_rebuilt_the_structure"],d["contacts_placed_without_rebuild"],d["contacts_created"]))
