cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rm -rf gpurun_out/prof4; mkdir -p gpurun_out/prof4
for L in 3 1; do
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/prof4/pmc1_L$L -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --lighting $L > /dev/null 2>&1 || true
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_ADD_F64 --output-format csv -d gpurun_out/prof4/pmc2_L$L -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --lighting $L > /dev/null 2>&1 || true
done
make -C all_is_cubes_amd/csrc clean >/dev/null; make -C all_is_cubes_amd/csrc HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DAIC_PROFILE" >/dev/null 2>&1
python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep PROF | tail -12
