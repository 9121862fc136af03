cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_repro tools/pk_f32_mfma_repro.hip 2>/dev/null && timeout 900 /tmp/pk_repro 20 > gpurun_out/pk_repro.txt 2>&1
grep -n "^H\|^    lane" gpurun_out/pk_repro.txt | cut -c1-200 | awk '/^[0-9]+:H/{print; c=0; next} {c++; if (c<=3) print}'
