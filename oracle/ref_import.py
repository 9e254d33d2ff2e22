"""TEST INFRASTRUCTURE ONLY.

Imports the UNMODIFIED reference package ``/root/reference/rave`` in this build
container through the stub packages under ``oracle/shims`` (SURVEY.md section 8c /
Appendix A).  Used only by ``oracle/make_golden.py`` (to produce the committed
fixtures under ``tests/golden``) and by the ``not gpu`` tests that pin the oracle
restatement (``oracle/rave_oracle.py``) against the reference when
``/root/reference`` is present.  ``/root/reference`` does not exist on the GPU
box: nothing reachable from ``bench.py``, ``smoke()`` or the ``-m gpu`` tests
imports this module.
"""
import os
import sys

REFERENCE_ROOT = os.environ.get("RAVE_REFERENCE_ROOT", "/root/reference")
_SHIMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shims")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "rave"))


def import_reference():
    """Returns the reference's ``rave`` package (imported once)."""
    if "rave" in sys.modules and getattr(sys.modules["rave"], "__graft_reference__", False):
        return sys.modules["rave"]
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    # never write __pycache__ into the read-only reference tree
    sys.dont_write_bytecode = True
    for p in (_SHIMS, REFERENCE_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)

    # scipy aliases: rave/pqmf.py:10 imports scipy.signal.kaiser and calls
    # firwin(..., nyq=np.pi) (rave/pqmf.py:69); both were removed after the pinned
    # scipy==1.10.0 (requirements.txt:10).  nyq=pi == fs=2*pi (same normalisation).
    import scipy.signal
    import scipy.signal.windows

    if not hasattr(scipy.signal, "kaiser"):
        scipy.signal.kaiser = scipy.signal.windows.kaiser
    _firwin = scipy.signal.firwin
    if not getattr(_firwin, "__graft_wrapped__", False):
        def firwin(*args, nyq=None, **kwargs):
            if nyq is not None:
                kwargs["fs"] = 2.0 * nyq
            return _firwin(*args, **kwargs)

        firwin.__graft_wrapped__ = True
        scipy.signal.firwin = firwin

    import gin

    gin.clear_bindings()
    # configs/v1.gin:33-34, :41
    gin.bind("cc.Conv1d", bias=False)
    gin.bind("cc.ConvTranspose1d", bias=False)
    gin.bind("normalization", mode="weight_norm")

    import rave  # noqa: E402  (the reference package)

    rave.__graft_reference__ = True
    return rave


def set_causal(flag: bool):
    """configs/causal.gin:5  ``cc.get_padding.mode = 'causal'`` (changes the default)."""
    import gin

    if flag:
        gin.bind("cc.get_padding", mode="causal")
    else:
        gin._BINDINGS.pop("cc.get_padding", None)
