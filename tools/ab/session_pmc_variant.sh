# round 6: the SQ counter passes of tools/pmc_c3_stalls.sh / pmc_fetch_levels.sh on a variant build: bash tools/ab/session_pmc_variant.sh <variant> <workload> <tag>
cd $GRAFT_REPO_ROOT
V=$1; W=${2:-c3}; O=$GRAFT_REPO_ROOT/gpurun_out/${3:-r06e}; mkdir -p $O
cp sponge_amd/libposeidon_mi355x.so /tmp/keep.so
cp tools/ab/libposeidon_$V.so sponge_amd/libposeidon_mi355x.so
python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$W $V %.4g perm/s  kernel_ms %.4f verified %s clk %.3g'%(d['value'],d['roofline']['kernel_ms'],d['verified'],d['int_valu']['shader_clock_hz']))" | tee $O/${V}_${W}_bench.txt
bash tools/pmc_c3_stalls.sh $W $O/stalls_${V}_$W 2>&1 | tee $O/${V}_${W}_stalls.txt
bash tools/pmc_fetch_levels.sh $W $O/levels_${V}_$W 2>&1 | tee $O/${V}_${W}_levels.txt
cp /tmp/keep.so sponge_amd/libposeidon_mi355x.so
