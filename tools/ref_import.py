"""Import the reference's hot-path classes in the DEV CONTAINER only.

`/root/reference` is read-only and does not exist on the GPU box; nothing in
tests/, bench.py or __graft_entry__.py imports this module.  It is used by
tools/make_golden.py to generate committed golden vectors.

The reference imports `future`, `past`, `pysam` and its unbuilt Cython shim at
module scope but none of them is used by TelescopeLikelihood / csr_matrix_plus
/ Telescope.save|load|output_report, so inert `sys.modules` entries suffice
(SURVEY.md section 8(c)).
"""
import sys
import types

REF_ROOT = '/root/reference'


def load_reference():
    sys.dont_write_bytecode = True

    def _mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    if 'future' not in sys.modules:
        fut = _mod('future')
        fut.standard_library = _mod('future.standard_library',
                                    install_aliases=lambda: None)
        past = _mod('past')
        past.utils = _mod('past.utils', old_div=lambda a, b: a // b
                          if isinstance(a, int) and isinstance(b, int) else a / b)
        _mod('pysam')
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import telescope.utils  # noqa: F401  (package import first)
    _mod('telescope.utils.calignment', AlignedPair=object)
    from telescope.utils.model import Telescope, TelescopeLikelihood
    from telescope.utils.sparse_plus import csr_matrix_plus
    return Telescope, TelescopeLikelihood, csr_matrix_plus
