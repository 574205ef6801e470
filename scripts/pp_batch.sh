b() { timeout 300 python bench.py --mode batch --frames 512 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['frames_per_s'], d['frames_per_s_each_round'])"; }
export HYDAMD_CONTEXT_CACHE=32 HYDAMD_RANS_WAVES=5
for lf in 1 2; do for t in 10 14 18; do
echo -n "form 5, LF coder mode $lf threads $t: "; HYDAMD_LF_CODER=$lf b --threads $t
done; done
