"""TEST INFRASTRUCTURE ONLY: CPU restatement (pure PyTorch fp32/fp64) of the reference's step path, used
by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline as the checker.  Never imported by the
product package vision_mtl_amd."""
