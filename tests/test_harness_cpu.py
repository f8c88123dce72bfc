"""-m "not gpu": the model harness graph (mocopci_amd.model) run on the CPU oracle backend against the
reference's stored outputs.  This pins the ORACLE and the harness against the reference's own Python."""
import pytest
import torch

from mocopci_amd import ops
from oracle.backend import OracleBackend
from oracle import pointset as orc
from tests import harness_checks as hc


@pytest.fixture()
def oracle_backend():
    prev = ops.set_backend(OracleBackend())
    yield
    ops.set_backend(prev)


def test_layers_match_reference(oracle_backend):
    hc.run_layer_checks("cpu")


def test_forward_config1_matches_reference(oracle_backend):
    hc.run_forward_check("cpu", "forward_c1_n1024", 1, 1, 1024, orc.chamfer)


def test_forward_batched_matches_reference(oracle_backend):
    hc.run_forward_check("cpu", "forward_b2_n2048", 6, 2, 2048, orc.chamfer)


def test_forward_baseline_point_count_matches_reference(oracle_backend):
    """BASELINE configs[1]'s point count (N=8192, sequence 0 of config 2) against the reference's own forward."""
    hc.run_forward_check("cpu", "forward_c2_n8192", 2, 1, 8192, orc.chamfer)


def test_state_dict_keys_match_reference_spec():
    import json, os
    spec = json.load(open(os.path.join(hc.GOLD, "state_dict_spec.json")))
    net = hc.build_model("cpu")
    sd = net.state_dict()
    assert list(sd.keys()) == list(spec.keys()) and len(sd) == 487
    assert sum(v.numel() for v in sd.values()) == 5945990
    for k, v in sd.items():
        assert list(v.shape) == spec[k]["shape"], k


def test_hip_backend_refuses_cpu_tensors():
    """No silent CPU fallback on the product path."""
    be = ops.HipBackend()
    with pytest.raises(RuntimeError):
        be.fps(torch.zeros(1, 16, 3), 4)
