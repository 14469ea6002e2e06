cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_unaligned_external.py tests/test_gpu_parity.py tests/test_algorithms.py tests/test_slices_centroid_views.py -m gpu -q -p no:cacheprovider -k "external or pinned or torch_tensors or refused or Columns or columns_buffer or slice" 2>&1 | tail -15 | cut -c1-400
