# same-box A/B of the weight-gradient kernel forms (profiles/r06_dw_split_notes.md): micro-benchmark (+ timing twins built with
# tools/build_objs.sh dww_X pnr_bwd=-DPNR_X_DWW_X, listed in $VARIANTS), gradient tests, training step
O=gpurun_out/${1:-r06_s46}; mkdir -p $O
(PNR_DW_FORM=8wave python tools/gpu_dw_bench.py; python tools/gpu_dw_bench.py; for v in $VARIANTS; do PIXELNERF_HIP_LIB=build/libpnr_$v.so PIXELNERF_ALLOW_VARIANT=1 python tools/gpu_dw_bench.py; done; PNR_DW_FORM=8wave python tools/gpu_dw_bench.py; python tools/gpu_dw_bench.py) 2>&1 | grep -v amdgpu.ids | tee $O/dw_bench.txt
(timeout 900 python -m pytest tests/test_hip_backward.py tests/test_hip_backward_f32.py tests/test_hip_trained_weights.py tests/test_hip_graph_capture.py tests/test_hip_sparse_fold.py -m gpu -x -q 2>&1 | tail -5) | tee $O/tests.txt
(for i in 1 2; do echo "== PNR_DW_FORM=8wave"; PNR_DW_FORM=8wave python tools/gpu_train_f16x3_quick.py train; echo "== one wave per SIMD (default)"; python tools/gpu_train_f16x3_quick.py train; done) 2>&1 | grep -v amdgpu.ids | tee $O/train_ab.txt
