import json, sys
d=json.load(open(sys.argv[1]))
c3=d["config3"]["generate_obj_mesh_wnf_t2d"]; c5=d["config5"]
print(sys.argv[1], "c3", round(c3["simple_local"]["end_to_end_ms"],3), round(c3["attention_local"]["end_to_end_ms"],3), "c5", round(c5["end_to_end_ms"],3))
