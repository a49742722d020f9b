#!/bin/bash
# per-layer times IN THE STEP'S SEQUENCE (tools/profile_layers.py) under tile / slicing knobs: which choice each layer's cost model should make
O=gpurun_out/sweep; mkdir -p $O
run() { name=$1; shift; env "$@" python3 tools/profile_layers.py 2>&1 | grep -v amdgpu > $O/$name.txt; }
run default RTN_X=0
run default2 RTN_X=0
run g8mi2 RTN_CONV_G8_MI=2
run g8mi3 RTN_CONV_G8_MI=3
run h8mi3 RTN_CONV_H8_MI=3
run h8mi4 RTN_CONV_H8_MI=4
run h8ks1 RTN_CONV_H8_KSPLIT=1
run h8ks2 RTN_CONV_H8_KSPLIT=2
run h8ks6 RTN_CONV_H8_KSPLIT=6
run g8off RTN_CONV_G8=0
run h8off RTN_CONV_H8=0
python3 - <<'PY'
import glob,os,re
O="gpurun_out/sweep"
names=["default","default2","g8mi2","g8mi3","h8mi3","h8mi4","h8ks1","h8ks2","h8ks6","g8off","h8off"]
T={}
for n in names:
    for l in open(os.path.join(O,n+".txt")):
        p=l.split()
        if len(p)>=2 and re.match(r"^\d+\.\d+$",p[1]): T.setdefault(p[0],{})[n]=float(p[1])
print("%-30s"%"op"+"".join("%9s"%n for n in names))
tot={n:0 for n in names}
for op,d in T.items():
    base=min(d.get("default",9),d.get("default2",9))
    row="%-30s"%op
    for n in names:
        v=d.get(n); 
        if v is None: row+="%9s"%"-"; continue
        mark="*" if v<base*0.96 else " "
        row+="%8.4f%s"%(v,mark)
    print(row)
PY
