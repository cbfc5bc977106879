"""The C-ABI library loads on a CPU-only machine and exports every symbol the header declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "miosqp_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(miosqp_qp_[a-z_]+)\s*\(", txt)))


def test_library_exports_header_symbols():
    from miosqp_amd import _lib
    lib = _lib.load()
    names = _header_functions()
    assert len(names) >= 16
    for nm in names:
        assert hasattr(lib, nm), nm
    assert sorted(_lib.SYMBOLS) == names


def test_constants_and_defaults_without_gpu():
    from miosqp_amd import qp
    assert qp.constant("OSQP_SOLVED") == 1
    assert qp.constant("OSQP_MAX_ITER_REACHED") == -2
    assert qp.constant("OSQP_PRIMAL_INFEASIBLE") == -3
    assert qp.constant("OSQP_DUAL_INFEASIBLE") == -4
    assert qp.constant("OSQP_UNSOLVED") == -10
    with pytest.raises(ValueError):
        qp.constant("OSQP_NOPE")
    s = qp.default_settings()
    assert (s.rho, s.sigma, s.alpha, s.max_iter, s.scaling, s.check_termination) == (0.1, 1e-6, 1.6, 4000, 10, 25)


def test_settings_validation_without_gpu():
    from miosqp_amd import qp
    with pytest.raises(ValueError):
        qp._settings_from_kwargs({"adaptive_rho": True})
    with pytest.raises(TypeError):
        qp._settings_from_kwargs({"no_such_setting": 1})
    s = qp._settings_from_kwargs({"eps_inf": 1e-5, "polishing": False, "verbose": False, "rho": 0.3})
    assert s.eps_prim_inf == 1e-5 and s.rho == 0.3


def test_no_silent_cpu_fallback():
    """Without a GPU the product must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    from miosqp_amd import bnb, problems
    pr = problems.random_miqp(10, 5, 2, seed=0)
    with pytest.raises(RuntimeError, match="no HIP device"):
        bnb.MIOSQP().setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"],
                           pr["i_u"], dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "miosqp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", ".inc", ".c", ".cc", ".txt")) or f == "Makefile":
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# tests pass the CPU oracle", "").lower() or \
                    f in ("bnb.py",), (dirpath, f)


def test_cooperative_exchange_loop_has_no_register_spills():
    """tools/check_coop_isa.py: the exchange loop of every k_coop instantiation (inline-asm loads whose completion the
    compiler cannot see): along every path of the poll loop, no instruction between such a load and its s_waitcnt may name
    the load's destination registers or touch scratch -- a spill or copy there would move a register whose data has not
    arrived.  Compiles the device code to assembly (hipcc cross-compiles here, ~15 s)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        import pytest
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_coop_isa.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr  # (no scratch access between a poll load and its wait, anywhere)
    assert r.stdout.count("that touch its registers (or scratch) 0") >= 12
    # the loop of the headline (config 2: 3 columns per thread, testers, the grid RESIDENT over a search_run call --
    # coop_grid_run, the function the resident kernel calls per node): spill-free
    # (and coop_grid_one, the same grid in a launch of its own: solve / solve_node / a node of a search on a called-off run)
    import re
    for fn, max_instr in (("coop_grid_runILi512ELi8ELi3ELi4E", 520), ("coop_grid_oneILi512ELi8ELi3ELi4E", 430)):
        head = [ln for ln in r.stdout.splitlines() if fn in ln]
        assert head and "scratch accesses 0," in head[0], head
        # ... within its register budget with room to spare (256 per lane at two waves per SIMD), and no longer than it was
        # when it was measured (r06: 482 / 397 instructions, 230 / 228 registers): a change to the loop shows here before
        # it shows on a GPU
        vg = int(re.search(r"vector registers of the function (\d+)", head[0]).group(1))
        ins = int(re.search(r"instructions in the loop (\d+)", head[0]).group(1))
        assert 0 < vg <= 240 and 0 < ins <= max_instr, (fn, vg, ins)


def test_every_environment_switch_is_documented():
    """SWITCHES.md lists every MIOSQP_* variable the library, the Python front and bench.py read -- and none that nobody
    reads (operability: 48 switches scattered over 16 files with no single place that says what they do)."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    read = set()
    csrc = os.path.join(root, "miosqp_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".inc", ".cpp", ".hpp")):
            read |= set(re.findall(r'getenv(?:_off)?\("(MIOSQP_[A-Z0-9_]+)"', open(os.path.join(csrc, f)).read()))
    for f in [os.path.join(root, "bench.py")] + [os.path.join(root, "miosqp_amd", g) for g in os.listdir(os.path.join(root, "miosqp_amd"))
                                                 if g.endswith(".py")]:
        read |= set(re.findall(r'environ(?:\.get)?[\[(]\s*"(MIOSQP_[A-Z0-9_]+)"', open(f).read()))
    doc = set(re.findall(r"`(MIOSQP_[A-Z0-9_]+)`", open(os.path.join(root, "SWITCHES.md")).read()))
    assert read - doc == set(), "read but not documented: %s" % sorted(read - doc)
    assert doc - read == set(), "documented but never read: %s" % sorted(doc - read)
