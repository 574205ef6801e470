for cs in 2 4 8; do for st in 2 4 8 16; do echo -n "copy streams $cs stage threads $st: "; HYDAMD_COPY_STREAMS=$cs HYDAMD_STAGE_THREADS=$st python scripts/batch_client.py 10 2>&1 | tail -1 | cut -c1-70; done; done
echo -n "eager off: "; HYDAMD_EAGER=0 python scripts/batch_client.py 10 2>&1 | tail -1 | cut -c1-70
echo -n "lf 0: "; HYDAMD_LF_CODER=0 python scripts/batch_client.py 10 2>&1 | tail -1 | cut -c1-70
echo -n "host assembly: "; HYDAMD_HOST_ASSEMBLY=1 python scripts/batch_client.py 10 2>&1 | tail -1 | cut -c1-70
