"""gflags-compatible parser: the reference's command line (README.md:30-53) must parse unchanged."""
import pytest
from hypothesis import given, strategies as st

from multimodalgame_amd import flags as F

README_ARGV = ["model.py", "-experiment_name", "demo", "-exchange_samples", "5", "-model_type", "Adaptive",
               "-max_exchange", "10", "-batch_size", "64", "-rec_w_dim", "32", "-sender_out_dim", "32",
               "-img_h_dim", "256", "-rec_hidden", "64", "-learning_rate", "1e-4", "-entropy_rec", "0.01",
               "-entropy_sen", "0.01", "-entropy_s", "0.08", "-use_binary", "-max_epoch", "500", "-top_k_dev", "6",
               "-top_k_train", "6", "-descr_train", "./utils/descriptions.csv", "-descr_dev", "./utils/descriptions.csv",
               "-train_file", "./utils/train.hdf5", "-dev_file", "./utils/dev.hdf5", "-wv_dim", "100",
               "-glove_path", "~/data/glove/glove.6B.100d.txt"]


@pytest.fixture(autouse=True)
def fresh():
    F.define_flags()
    F.FLAGS.Reset()
    yield
    F.FLAGS.Reset()


def test_defaults_match_reference():
    d = F.FLAGS.FlagValuesDict()
    assert d["batch_size"] == 32 and d["max_exchange"] == 3 and d["fixed_exchange"] is True
    assert d["optim_type"] == "RMSprop" and d["learning_rate"] == 1e-4 and d["entropy_s"] is None
    assert d["img_feat_dim"] == 4096 and d["rec_hidden"] == 128 and d["baseline_hid_dim"] == 500
    additive = ("seed", "max_steps", "synthetic_data", "world_size", "rank", "dist_backend")      # "additive only"
    assert len([k for k in d if k not in additive]) == 74


def test_readme_command_line():
    F.FLAGS(README_ARGV)
    F.default_flags(README_ARGV)
    f = F.FLAGS
    assert f.model_type == "Adaptive" and f.fixed_exchange is False and f.img_feat_dim == 512
    assert f.max_exchange == 10 and f.batch_size == 64 and f.rec_w_dim == 32 and f.img_h_dim == 256
    assert f.entropy_s == 0.08 and f.use_binary is True and f.exchange_samples == 5
    assert f.log_file == "./logs/demo.log" and f.checkpoint == "./logs/demo.pt" and f.json_file == "./logs/demo.json"
    assert f.binary_output == "./logs/demo.bv.hdf5" and not f.glove_path.startswith("~")


def test_command_line_overrides_preset():
    argv = ["model.py", "-model_type", "Adaptive", "-fixed_exchange"]
    F.FLAGS(argv)
    F.default_flags(argv)
    assert F.FLAGS.fixed_exchange is True        # re-parse after the preset (model.py:1752-1754)


def test_boolean_forms_enum_validation_and_unknown():
    F.FLAGS(["x", "-nouse_binary", "--cuda", "-shuffle_train=false", "-max_exchange=7"])
    assert F.FLAGS.use_binary is False and F.FLAGS.cuda is True and F.FLAGS.shuffle_train is False
    assert F.FLAGS.max_exchange == 7
    with pytest.raises(F.FlagsError):
        F.FLAGS(["x", "-optim_type", "Adagrad"])
    with pytest.raises(F.FlagsError):
        F.FLAGS(["x", "-no_such_flag", "1"])
    assert F.FLAGS(["x", "pos", "-debug"]) == ["x", "pos"]


def test_log_load_roundtrip(tmp_path):
    F.FLAGS(["x", "-max_exchange", "9", "-nouse_binary", "-experiment_name", "a"])
    p = tmp_path / "a.json"
    p.write_text(__import__("json").dumps(F.FLAGS.FlagValuesDict()))
    F.FLAGS.Reset()
    argv = ["x", "-log_load", str(p), "-experiment_name", "b"]
    F.FLAGS(argv)
    F.default_flags(argv)
    assert F.FLAGS.max_exchange == 9 and F.FLAGS.use_binary is False and F.FLAGS.experiment_name == "b"


@given(st.integers(min_value=-10**6, max_value=10**6), st.booleans())
def test_integer_and_bool_roundtrip(n, b):
    F.define_flags()
    F.FLAGS(["x", "-save_after", str(n), "-debug" if b else "-nodebug"])
    assert F.FLAGS.save_after == n and F.FLAGS.debug is b


@pytest.mark.parametrize("argv", [["-desc_attn"], ["-sender_mix", "prod"], ["-sender_mix", "mou"], ["-flipout_sen", "0.1"],
                                  ["-flipout_rec", "0.05"], ["-ignore_receiver"], ["-ignore_code"], ["-visual_attn"],
                                  ["-bit_flip"]])
def test_unsupported_reference_switches_raise(argv):
    """Flags the reference reads at model.py:201-221, 233-234, 344, 467-470, 813 parse but must not be silently ignored."""
    F.FLAGS(["model.py"] + argv)
    with pytest.raises(NotImplementedError) as e:
        F.check_supported()
    assert argv[0].lstrip("-") in str(e.value)


def test_supported_command_line_passes_the_check():
    F.FLAGS(README_ARGV)
    F.default_flags(README_ARGV)
    F.check_supported()


def test_game_constructor_checks_flags():
    """Game.__init__ is the choke point of every entry (exchange(), train_step, the CLI)."""
    from multimodalgame_amd.game import Game

    class Fl(object):
        desc_attn, sender_mix, flipout_sen, flipout_rec = False, "sum", None, 0.1
        ignore_receiver = ignore_code = visual_attn = bit_flip = False
    with pytest.raises(NotImplementedError):
        Game(None, None, None, None, flags=Fl())
