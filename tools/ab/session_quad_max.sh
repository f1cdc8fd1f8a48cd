R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; cp sponge_amd/libposeidon_mi355x.so /tmp/orig.so
for round in 1 2; do for v in head q16k; do cp tools/ab/libposeidon_$v.so sponge_amd/libposeidon_mi355x.so; echo "== $v round $round"
for n in 20000 24576 32768; do for w in c2 h3 d3 k3; do python bench.py --workload $w --total-units $n --steps 50 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w $n: %.4e'%d['value'], '%.4f ms'%d['ms_per_step'], d['engine']['name'])"; done; done
python tools/merkle_levels.py 21 2>/dev/null | grep -E "2\^(15|16|17|21) leaves"
done; done
cp /tmp/orig.so sponge_amd/libposeidon_mi355x.so
