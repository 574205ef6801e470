B=scripts/probe_build
for lib in base nog; do
    echo "== $lib"; HYDAMD_LIB=$PWD/$B/k1v_$lib.so timeout 300 python scripts/pipe_probe.py --reps 3 --profile 0 2>&1 | tail -3
done
python scripts/host_cost.py 2>&1 | tail -5
