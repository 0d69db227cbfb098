"""TEST INFRASTRUCTURE — the CPU oracle.

`oracle.pointops_ref`  ctypes front-end of liboracle.so (C restatement of the reference's
                       pointops2 kernels, oracle/pointops_oracle.c)
`oracle.index_ref`     numpy/torch-CPU restatement of the reference's index build
                       (grid_sample / get_indice_pairs / CSR / rel-pos index)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this package.
The product (stratified_transformer_amd/) never does.
"""
