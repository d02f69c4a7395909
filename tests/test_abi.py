"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/ctcdecode_amd.h
declares (no compute calls -- there is no GPU here), and the Python class mirrors the reference's signature."""
import inspect
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "ctcdecode_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ctcd_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g

    g.build()
    import ctypes

    from ctcdecode_amd import _build

    lib = ctypes.CDLL(_build.LIB_PATH)
    names = _declared()
    assert "ctcd_beam_decode" in names and "ctcd_beam_decode_host" in names and len(names) >= 9
    for name in names:
        assert hasattr(lib, name), name
    lib.ctcd_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.ctcd_version()
    lib.ctcd_workgroup_lds_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    assert 0 < lib.ctcd_workgroup_lds_bytes(100, 29, 40, 1.0) <= 160 * 1024  # BASELINE.json configs[1] fits one CU's LDS


def test_python_signature_mirrors_reference():
    import ctcdecode_amd

    sig = inspect.signature(ctcdecode_amd.CTCBeamDecoder.__init__)
    names = list(sig.parameters)[1:11]
    # ctcdecode/__init__.py:26-38 of the reference
    assert names == ["labels", "model_path", "alpha", "beta", "cutoff_top_n", "cutoff_prob", "beam_width", "num_processes", "blank_id", "log_probs_input"]
    defaults = {k: v.default for k, v in sig.parameters.items() if v.default is not inspect._empty}
    assert defaults["cutoff_top_n"] == 40 and defaults["cutoff_prob"] == 1.0 and defaults["beam_width"] == 100
    assert defaults["num_processes"] == 4 and defaults["blank_id"] == 0 and defaults["log_probs_input"] is False
    assert list(inspect.signature(ctcdecode_amd.CTCBeamDecoder.decode).parameters) == ["self", "probs", "seq_lens"]


def test_online_signature_mirrors_reference():
    import ctcdecode_amd

    sig = inspect.signature(ctcdecode_amd.OnlineCTCBeamDecoder.__init__)
    assert list(sig.parameters)[1:11] == ["labels", "model_path", "alpha", "beta", "cutoff_top_n", "cutoff_prob", "beam_width", "num_processes", "blank_id", "log_probs_input"]
    # ctcdecode/__init__.py:189
    dsig = inspect.signature(ctcdecode_amd.OnlineCTCBeamDecoder.decode)
    assert list(dsig.parameters)[:5] == ["self", "probs", "states", "is_eos_s", "seq_lens"]
    # (extensions come after the reference's parameters and have defaults that keep the reference's behaviour: check=True)
    assert all(p.default is not inspect._empty for p in list(dsig.parameters.values())[5:]) and dsig.parameters["check"].default is True
    assert list(inspect.signature(ctcdecode_amd.DecoderState.__init__).parameters) == ["self", "decoder"]


def test_product_does_not_touch_the_oracle():
    """The product path must never import/load anything under oracle/ (it is test infrastructure)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ctcdecode_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle/" not in text.replace("oracle/ (", "").replace("under oracle/", "") or f == "__init__.py", f
                assert "libctcoracle" not in text and "libctcref" not in text, f


def test_reference_import_name():
    """README.md:22-38 of the reference: ``from ctcdecode import CTCBeamDecoder`` (+ the online classes) works unchanged."""
    import ctcdecode
    import ctcdecode_amd

    assert ctcdecode.CTCBeamDecoder is ctcdecode_amd.CTCBeamDecoder
    assert ctcdecode.OnlineCTCBeamDecoder is ctcdecode_amd.OnlineCTCBeamDecoder and ctcdecode.DecoderState is ctcdecode_amd.DecoderState
