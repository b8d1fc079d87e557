"""CPU tests of small host-side pieces added in round 4 (no GPU, no library calls that compute)."""
import types

import torch

from dgcnn_amd import dist as ddist


def test_device_identity_prefers_uuid_then_pci_bus_id_never_the_ordinal(monkeypatch):
    """ranks launched with their own HIP_VISIBLE_DEVICES all see ordinal 0: the same-device test of the one-shot exchange must
    compare physical identities"""
    props = types.SimpleNamespace(uuid="GPU-1234", pci_bus_id=7, pci_domain_id=0)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: props)
    assert ddist._device_identity(torch.device("cuda", 0)) == "uuid:GPU-1234"
    props2 = types.SimpleNamespace(pci_bus_id=7, pci_domain_id=1)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: props2)
    a = ddist._device_identity(torch.device("cuda", 0))
    assert a == "pci_bus_id:1:7"
    props3 = types.SimpleNamespace(pci_bus_id=9, pci_domain_id=1)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: props3)
    assert ddist._device_identity(torch.device("cuda", 0)) != a          # same ordinal, another bus: another device

    def boom(d):
        raise RuntimeError("no device")
    monkeypatch.setattr(torch.cuda, "get_device_properties", boom)
    u = ddist._device_identity(torch.device("cuda", 0))
    assert u.startswith("unknown:")                                      # no identity: never equal across processes


def test_step_kernel_switch_is_exported_and_reports_its_previous_value():
    from dgcnn_amd import _lib
    L = _lib.lib()
    prev = L.dgcnn_step_kernel_enable(0)
    try:
        assert prev in (0, 1)
        assert L.dgcnn_step_kernel_enable(1) == 0
        # the form bit follows the switch (pure host function)
        f = L.dgcnn_forward_form
        CU = _lib.FLAG_COALESCED_UNDIRECTED
        assert f(3800, 140000, 50, 1, CU, 180) & 8
        L.dgcnn_step_kernel_enable(0)
        assert not f(3800, 140000, 50, 1, CU, 180) & 8
    finally:
        L.dgcnn_step_kernel_enable(prev)
