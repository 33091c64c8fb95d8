"""CPU-side checks of the drop-in boundary: libmmg.so loads, exports every symbol include/mmg.h
declares, and its layout queries (no GPU needed) agree with the reference's parameter inventory."""
import ctypes as C
import os
import re

from multimodalgame_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "mmg.h")).read()
    declared = set(re.findall(r"\b(mmg_[a-z_]+)\s*\(", header))
    declared -= {"mmg_config", "mmg_handle"}
    lib = _lib.load()
    for sym in sorted(declared):
        assert hasattr(lib, sym), "libmmg.so does not export " + sym
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    assert lib.mmg_version() == 3


def test_param_table_matches_reference_inventory():
    # config 1: 384 180 parameters; state_dict keys of SURVEY.md §8(b)
    cfg = _lib.make_config(64, 30, 512, 256, 32, 64, 100, 500, 10, fixed_exchange=False)
    tab = _lib.param_table(cfg)
    assert sum(e["rows"] * max(e["cols"], 1) for e in tab) == 384180
    by_agent = {}
    for e in tab:
        by_agent.setdefault(e["agent"], []).append(e["name"])
        assert e["offset"] % 4 == 0
    assert by_agent["sender"] == ["image_layer.weight", "image_layer.bias", "code_layer.weight", "code_layer.bias",
                                  "code_bias", "binary_layer.weight", "binary_layer.bias"]
    assert set(by_agent["receiver"]) == {"rnn.weight_ih", "rnn.weight_hh", "rnn.bias_ih", "rnn.bias_hh", "w_h.weight",
                                         "w_h.bias", "w_d.weight", "w.weight", "w.bias", "y1.weight", "y1.bias",
                                         "y2.weight", "y2.bias", "s.weight", "s.bias"}
    assert by_agent["baseline_rec"] == by_agent["baseline_sen"] == ["linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias"]
    # config 4: 1 952 852 parameters
    cfg4 = _lib.make_config(64, 30, 512, 1024, 256, 64, 100, 500, 10, fixed_exchange=False)
    assert sum(e["rows"] * max(e["cols"], 1) for e in _lib.param_table(cfg4)) == 1952852


def test_bad_config_reports_an_error():
    lib = _lib.load()
    cfg = _lib.make_config(0, 30, 512, 256, 32, 64, 100, 500, 10)
    assert lib.mmg_param_count(C.byref(cfg)) < 0
    assert b"positive" in lib.mmg_last_error()


def test_tape_table_is_consistent():
    cfg = _lib.make_config(64, 30, 512, 256, 32, 64, 100, 500, 10, fixed_exchange=False)
    tab = _lib.tape_table(cfg)
    names = [e["name"] for e in tab]
    assert len(set(names)) == len(names)
    for k in ("y", "z", "pz", "w", "pw", "s", "ps", "mask", "bs", "br", "losses", "stats"):
        assert k in names
    end = 0
    for e in tab:
        assert e["offset"] >= end and e["offset"] % 256 == 0
        end = e["offset"]
    assert _lib.load().mmg_workspace_bytes(C.byref(cfg)) > end


def test_documents_name_only_symbols_that_exist():
    """Every mmg_* token of INTEGRATION.md / DESIGN.md / README.md is an exported entry point (or a type of the header), and
    every class / method the integration guide tells a maintainer to call exists."""
    header_types = {"mmg_config", "mmg_handle", "mmg_param_entry", "mmg_tape_entry"}
    for doc in ("INTEGRATION.md", "DESIGN.md", "README.md"):
        text = open(os.path.join(REPO, doc)).read()
        for tok in sorted(set(re.findall(r"\bmmg_[a-z_]+\b", text))):
            assert tok in _lib.SYMBOLS or tok in header_types or tok in ("mmg_cli_", "mmg_minibatch_counter", "mmg_data"), "%s names %s, which libmmg.so does not export" % (doc, tok)
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    from multimodalgame_amd import agents, flags, game, misc
    for mod, names in ((game, re.findall(r"\bGame\.([a-z_]+)\(", text)), (flags, ["check_supported", "UNSUPPORTED", "define_flags", "default_flags"]),
                       (agents, ["Sender", "Receiver", "Baseline"]), (misc, ["_shuffled_order"])):
        for n in names:
            owner = game.Game if mod is game else mod
            assert hasattr(owner, n), "INTEGRATION.md names %s.%s, which does not exist" % (owner.__name__, n)
    assert "Trainer" not in text
