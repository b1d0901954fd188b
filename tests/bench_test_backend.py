"""The injected compute backend of `bench.py --test-backend bench_test_backend:make` (tests only): the CPU oracle stands in for the two HIP calls so that
bench.py's own launcher (`--gpus N` starting N ranks), its collective (gloo here, RCCL on the GPU box), the id globalisation + merge contract and the
compact headline run on CPU at world 2.  Lives under tests/: bench.py itself never imports the oracle for this."""
import numpy as np
import torch

import oracle_ffi as O
from test_sharded_gloo import OracleBackend


def make(rank, world, row_seed, row0, n, dim, nqueries, query_seed):
    rows = O.preprocess(O.COSINE, O.synth(row_seed, row0, n, dim))
    queries = torch.from_numpy(np.ascontiguousarray(O.synth(query_seed, 0, nqueries, dim)))
    return OracleBackend(O, rows, O.COSINE), queries
