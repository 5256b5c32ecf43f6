"""pysparse_amd.precon -- counterpart of pysparse.precon (pysparse/precon/__init__.py:7-17): the `precon`
extension module, plus the package-level `jacobi` / `ssor`, which the reference keeps as deprecated
forwarders to `precon.jacobi` / `precon.ssor` (they warn and call through)."""
import warnings as _warnings

from . import precon  # noqa: F401


def _deprecated(name):
    def forward(*args, **kwargs):
        # pysparse/misc/__init__.py: the Deprecated decorator warns with stacklevel 2, then calls through
        _warnings.warn("Call to deprecated method %r. Use pysparse.precon.precon.%s instead." % (name, name),
                       category=DeprecationWarning, stacklevel=2)
        return getattr(precon, name)(*args, **kwargs)
    forward.__name__ = name
    forward.__doc__ = getattr(precon, name).__doc__
    return forward


jacobi = _deprecated("jacobi")
ssor = _deprecated("ssor")
