set -e
export TMPDIR=/tmp
OUT=gpurun_out/sched_traffic; rm -rf $OUT; mkdir -p $OUT
for S in 0 1; do
  for B in 262144; do
    python3 tools/launch_config.py --horizon 20 --dtype f32 --batch $B --cfg 3 --steps 5 --schedule $S | tail -1
    python3 tools/launch_config.py --horizon 20 --dtype f64 --batch $B --cfg 4 --steps 3 --schedule $S | tail -1
    rocprofv3 --pmc FETCH_SIZE -d $OUT/f$S -o c -- python3 tools/launch_config.py --horizon 20 --dtype f32 --batch $B --cfg 3 --steps 3 --schedule $S > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE -d $OUT/w$S -o c -- python3 tools/launch_config.py --horizon 20 --dtype f32 --batch $B --cfg 3 --steps 3 --schedule $S > /dev/null 2>&1
  done
done
python3 - <<'PY'
import sqlite3, glob
for S in (0,1):
    v={}
    for sub,c in (("f","FETCH_SIZE"),("w","WRITE_SIZE")):
        db=glob.glob("gpurun_out/sched_traffic/%s%d/**/*_results.db"%(sub,S),recursive=True)[0]
        r=sqlite3.connect(db).execute("select avg(value) from counters_collection where counter_name=? and kernel_name like '%kmpc_solve_fast_kernel<float, 20>%'",(c,)).fetchone()
        v[c]=r[0]
    print("schedule",S,v,"HBM MB per dispatch %.1f"%((2*v["FETCH_SIZE"]+v["WRITE_SIZE"])*1024/1e6))
PY
rm -rf $OUT
