cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/enc; rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/enc -- python3 $R/tools/profile_encoder.py 2>&1 | grep encode_music
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
rows=[]
for f in glob.glob(R+"/gpurun_out/enc/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Grid_Size_X"], r["Workgroup_Size_X"]))
rows.sort()
# last third pass: find k_me kernels of the last encode
me=[r for r in rows if "k_me_" in r[1]]
n=len(me)//3
for r in me[2*n:]:
    print(f"{r[2]:9.1f} us  grid {r[3]:>9s}  {r[1][:60]}")
print("total", sum(r[2] for r in me[2*n:]))
PY
