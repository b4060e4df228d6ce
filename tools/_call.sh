cd $GRAFT_REPO_ROOT; export HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2 3; do python -m pytest "tests/test_gpu_allreduce.py::test_custom_allreduce_processes_on_one_gpu" -m gpu -q -x -k "kernels-8" 2>&1 | grep -v "^\[W\|amdgpu.ids\|Gloo" | tail -40; done
