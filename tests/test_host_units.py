"""CPU tests of small host-side pieces added in round 4 (no GPU, no library calls that compute)."""
import types

import torch

from dgcnn_amd import dist as ddist


def test_device_identity_combines_uuid_and_pci_address_never_the_ordinal(monkeypatch):
    """ranks launched with their own HIP_VISIBLE_DEVICES all see ordinal 0: the same-device test of the one-shot exchange must
    compare physical identities; an all-zero uuid is not an identity and the PCI address is always part of it
    (builds that report one constant uuid for every GPU)"""
    dev = torch.device("cuda", 0)
    props = types.SimpleNamespace(uuid="GPU-1234", pci_bus_id=7, pci_domain_id=0, pci_device_id=0)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: props)
    a0 = ddist._device_identity(dev)
    assert "uuid:GPU-1234" in a0 and "pci:0:7:0" in a0
    props_b = types.SimpleNamespace(uuid="GPU-1234", pci_bus_id=9, pci_domain_id=0, pci_device_id=0)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: props_b)
    assert ddist._device_identity(dev) != a0                              # one constant uuid, another bus: another device
    zero = types.SimpleNamespace(uuid="00000000-0000-0000-0000-000000000000", pci_bus_id=7, pci_domain_id=1, pci_device_id=0)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: zero)
    z = ddist._device_identity(dev)
    assert "uuid" not in z and z == "pci:1:7:0"
    props2 = types.SimpleNamespace(pci_bus_id=7, pci_domain_id=1)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: props2)
    a = ddist._device_identity(dev)
    assert a == "pci:1:7:0"
    props3 = types.SimpleNamespace(pci_bus_id=9, pci_domain_id=1)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: props3)
    assert ddist._device_identity(dev) != a                               # same ordinal, another bus: another device
    nothing = types.SimpleNamespace(uuid="00000000-0000-0000-0000-000000000000")
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: nothing)
    assert ddist._device_identity(dev).startswith("unknown:")            # zero uuid and no PCI address: no identity

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


def test_flat_adam_without_torchs_step_wrapper_still_runs_hooks_schedulers_and_the_fallback():
    """dgcnn_amd.optim.Adam marks its ``step`` as hooked (no profiler range / hook dispatch per call); step hooks registered on
    the optimizer or globally, LR schedulers and ``zero_grad`` must behave as with torch.optim.Adam -- checked on CPU
    parameters, i.e. through the fallback to torch's own Adam (same numbers)"""
    import torch
    from torch.optim.optimizer import register_optimizer_step_post_hook
    from dgcnn_amd.optim import Adam as FlatAdam
    torch.manual_seed(3)
    a, b = torch.nn.Linear(5, 3), torch.nn.Linear(5, 3)
    b.load_state_dict(a.state_dict())
    oa, ob = FlatAdam(a.parameters(), lr=1e-2), torch.optim.Adam(b.parameters(), lr=1e-2)
    assert getattr(type(oa).step, "hooked", False) and type(oa).step.__name__ == "step"      # torch installed no wrapper
    sa, sb = torch.optim.lr_scheduler.StepLR(oa, 2, 0.5), torch.optim.lr_scheduler.StepLR(ob, 2, 0.5)
    calls = []
    h1 = oa.register_step_post_hook(lambda opt, args, kwargs: calls.append("post"))
    h2 = oa.register_step_pre_hook(lambda opt, args, kwargs: calls.append("pre"))
    x = torch.randn(7, 5)
    for it in range(5):
        if it == 3:
            h1.remove(); h2.remove()
            h3 = register_optimizer_step_post_hook(lambda opt, args, kwargs: calls.append("global"))
        for m, o, s in ((a, oa, sa), (b, ob, sb)):
            m(x).square().sum().backward()
            o.step(); o.zero_grad(); s.step()
            assert all(p.grad is None for p in m.parameters())
    h3.remove()
    assert calls[:6] == ["pre", "post"] * 3 and calls.count("global") >= 2
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.equal(p, q)
    assert oa.param_groups[0]["lr"] == ob.param_groups[0]["lr"] == 1e-2 * 0.25
    oa.zero_grad(set_to_none=False)


def test_isa_audit_tool_reads_the_built_objects():
    """tools/isa_audit.py (the audit that found the sunk / eagerly selected loads and the scalar re-reads)
    disassembles the in-tree objects and reports the two patterns per kernel; the weight-gradient kernel's segment search is ONE
    batch of scalar loads since that audit: no scalar load of the <false> instantiation's main path sits in a loop any more"""
    import glob, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not glob.glob(os.path.join(root, "dgcnn_amd", "csrc", "*.o")) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        import pytest
        pytest.skip("no built objects / no llvm-objdump here")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "isa_audit.py"), "k_lin_first32s|k_gcn_fwd32n"],
                         capture_output=True, text=True, timeout=300).stdout
    lines = [l for l in out.splitlines() if "instr" in l]
    assert any("k_lin_first32s" in l for l in lines) and any("k_gcn_fwd32n" in l for l in lines), out
    for l in lines:
        if "k_gcn_fwd32n" in l:      # the narrow forward: no kernel-argument load behind its barrier
            assert l.rstrip().endswith("s_load behind a barrier   0"), l


def test_trainer_promises_an_exclusive_device_only_where_it_can_know():
    """DGCNN_FLAG_EXCLUSIVE_DEVICE (round 5): the in-launch wait of the fused preparation is admitted by a promise of the caller.
    Only the caller can make it (another process on the same GPU is invisible to this one): the default is no promise, an
    explicit True makes it, and the flag travels in the step arguments' flags word next to the batch's layout promise -- it
    never changes which kernel family a batch takes."""
    from dgcnn_amd import _lib
    from dgcnn_amd.model import Model
    from dgcnn_amd.train import Trainer
    m = Model(1, 3)
    assert Trainer(m)._excl == 0
    assert Trainer(m, exclusive_device=False)._excl == 0
    assert Trainer(m, exclusive_device=True)._excl == _lib.FLAG_EXCLUSIVE_DEVICE
    f = _lib.lib().dgcnn_forward_form
    CU = _lib.FLAG_COALESCED_UNDIRECTED
    assert f(3800, 140000, 50, 1, CU, 180) == f(3800, 140000, 50, 1, CU | _lib.FLAG_EXCLUSIVE_DEVICE, 180)
    assert f(9600, 360000, 128, 1, CU, 200) == f(9600, 360000, 128, 1, CU | _lib.FLAG_EXCLUSIVE_DEVICE, 200)


def test_data_can_restrict_its_own_forms():
    """``mode_flags`` on a batch (a PreparedDataset built without bitmap rows hands them out) reach the library's flags through both
    routes, Model.forward and the Trainer's cached step arguments"""
    import types
    from dgcnn_amd import _lib
    from dgcnn_amd.model import Model
    m = Model(1, 3)
    d = types.SimpleNamespace(coalesced_undirected=True, mode_flags=_lib.FLAG_AGG_SPARSE | _lib.FLAG_NO_CHAIN)
    fl = m._flags_of(d)
    assert fl & _lib.FLAG_COALESCED_UNDIRECTED and fl & _lib.FLAG_AGG_SPARSE and fl & _lib.FLAG_NO_CHAIN
    assert _lib.lib().dgcnn_forward_form(3800, 140000, 50, 1, fl, 180) == 0          # CSR gather kernels, whatever the sizes admit
