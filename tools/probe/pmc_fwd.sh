cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pp; timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pp -o x -- python tools/probe/fwd_stamps.py gps_slam_amd/libgpsslam_hip.so --no-stamps > /tmp/pp.log 2>&1
  python - "$grp" <<'P'
import csv,glob,sys,collections
acc=collections.defaultdict(list)
for fn in glob.glob('/tmp/pp/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'raster_ges_fwd_pk_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, 'n=%d'%len(v), 'last=%.0f'%v[-1])
if not acc: print('no rows for', sys.argv[1]); print(open('/tmp/pp.log').read()[-600:])
P
done
