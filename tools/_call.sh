cd $GRAFT_REPO_ROOT; export ROUND=r06 HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out/r06; mkdir -p $O
( python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py -m gpu -q -x 2>&1 | grep -v "^\[W\|amdgpu.ids\|Gloo" | tail -5 )
R=$O/attn_softmax_fma_ab.txt; : > $R
for rep in 1 2; do
for args in "--batch 64 --ctx 1024" "--batch 64 --ctx 4096 --int8" "--batch 16 --ctx 1024" "--batch 64 --ctx 1024 --int8" "--batch 8 --ctx 4096 --int8"; do
  echo "old: $(python tools/attn_bench.py $args 2>/dev/null | tail -1)" >> $R
  echo "new: $(python tools/attn_bench.py $args --product 2>/dev/null | tail -1)" >> $R
done; done
cat $R
