"""`DeepPolyComposer` (src/composer.rs:17-188) over the device entry points ms_horner_eval /
ms_deep_compose.  Same constructor arguments and method names; polynomials are coefficient-form
`Matrix` objects on the device (what `interpolate` / `into_polynomials` return)."""
import ctypes

import numpy as np

from .api import (FIELD_WORDS, GOLDILOCKS_FP, GOLDILOCKS_FQ3, STARK252_FP, GL_P, F252_P, GpuVec, Radix2EvaluationDomain, gl_from_mont, gl_to_mont,
                  f252_from_mont_limbs, f252_to_mont_limbs)


def _q_mul_base(q, s, p=GL_P):      # Fq element (canonical tuple or int) times canonical Fp scalar
    return tuple((c * s) % p for c in q) if isinstance(q, tuple) else (q * s) % p


def _q_pow(q, e, p=GL_P):
    """Fq3 = Fp[x]/(x^3 - 2) power on canonical tuples (host bookkeeping of points only)."""
    if not isinstance(q, tuple):
        return pow(q, e, p)

    def mul(a, b):
        a0, a1, a2 = a
        b0, b1, b2 = b
        return ((a0 * b0 + 2 * (a1 * b2 + a2 * b1)) % GL_P, (a0 * b1 + a1 * b0 + 2 * a2 * b2) % GL_P, (a0 * b2 + a1 * b1 + a2 * b0) % GL_P)
    r, base = (1, 0, 0), q
    while e:
        if e & 1:
            r = mul(r, base)
        base = mul(base, base)
        e >>= 1
    return r


def _words(q, field=GOLDILOCKS_FP):
    if field == STARK252_FP:
        return [int(w) for w in f252_to_mont_limbs(q % F252_P)]
    return [gl_to_mont(c) for c in q] if isinstance(q, tuple) else [gl_to_mont(q)]


def _from_words(w):
    if len(w) == 4:
        return f252_from_mont_limbs(w)
    return tuple(gl_from_mont(int(x)) for x in w) if len(w) == 3 else gl_from_mont(int(w[0]))


class _Cols:
    """stands in for a polynomial matrix where only its column count matters"""
    def __init__(self, k):
        self.k = k

    def num_cols(self):
        return self.k


class DeepCompositionCoeffs:                      # src/composer.rs:191-198
    def __init__(self, execution_trace, composition_trace, degree):
        self.execution_trace, self.composition_trace, self.degree = execution_trace, composition_trace, degree


_HORNER_MAX_COLS = 96          # msdeep::MAXCOLS (csrc/deep_kernels.h): columns per ms_horner_eval call


class DeepPolyComposer:
    """z and all coefficients are canonical Fq values: 3-tuples (Fq3) or ints (Fq = Fp AIRs, over Goldilocks
    or -- when the polynomial matrices are over it -- the 252-bit field).
    trace_arguments: the AIR's list of (column, offset) pairs (air.trace_arguments())."""

    def __init__(self, trace_arguments, trace_len, z, base_trace_polys, extension_trace_polys, composition_trace_polys):
        self.args = list(trace_arguments)
        self.z = z
        self.base = base_trace_polys
        self.ext = extension_trace_polys
        self.comp = composition_trace_polys
        self.n = trace_len
        self.planner = base_trace_polys.planner
        self.base_field = base_trace_polys.field
        self.p = F252_P if self.base_field == STARK252_FP else GL_P
        self.fq = GOLDILOCKS_FQ3 if isinstance(z, tuple) else self.base_field
        d = Radix2EvaluationDomain(trace_len, 1, self.base_field)
        self.g, self.g_inv = d.group_gen, d.group_gen_inv
        self.nbase = base_trace_polys.num_cols()
        self.next = extension_trace_polys.num_cols() if extension_trace_polys is not None else 0
        self._ood = None

    @classmethod
    def for_row_shards(cls, trace_arguments, trace_len, z, planner, nbase, next_, ncomp, ood, base_field=GOLDILOCKS_FP):
        """The composer of a rank of the multi-GPU prover (ministark_amd.distributed.prove_sharded): no rank holds every polynomial --
        the out-of-domain evaluations `ood` = (execution, composition) were computed by the columns' owners and gathered -- and
        into_deep_evaluations works on the rank's rows of the committed LDEs.  Columns are numbered as on one device."""
        self = cls.__new__(cls)
        self.args, self.z, self.n, self.planner, self.base_field = list(trace_arguments), z, trace_len, planner, base_field
        self.base = self.ext = None
        self.comp = _Cols(ncomp)
        self.p = F252_P if base_field == STARK252_FP else GL_P
        self.fq = GOLDILOCKS_FQ3 if isinstance(z, tuple) else base_field
        d = Radix2EvaluationDomain(trace_len, 1, base_field)
        self.g, self.g_inv = d.group_gen, d.group_gen_inv
        self.nbase, self.next, self._ood = nbase, next_, ood
        return self

    def _point(self, offset):
        gen = self.g if offset >= 0 else self.g_inv
        return _q_mul_base(self.z, pow(gen, abs(offset), self.p), self.p)

    def _horner(self, parts):
        """parts: list of (matrix, field, [(column, point), ...]) -> one list of Fq values per part.  Matrices over the same field with the same
        number of rows share ONE launch and one download (an Fq = Fp AIR's trace and composition-trace polynomials: every call is a wait of the
        host for the device and of the device for the host's next launch, ~75 us of an 8 ms proof)."""
        pl, L = self.planner, self.planner.lib
        pw = FIELD_WORDS[self.fq]
        results = [[] for _ in parts]
        buckets = {}
        for at, (matrix, field, queries) in enumerate(parts):
            if not queries:
                continue
            if matrix.field != field:
                raise ValueError("polynomial matrix is not over the expected field")
            key, k = (field, matrix.num_rows()), 0
            while sum(parts[m][0].num_cols() for m in buckets.get(key + (k,), [])) + matrix.num_cols() > _HORNER_MAX_COLS and buckets.get(key + (k,)):
                k += 1                                           # a launch takes at most msdeep::MAXCOLS columns
            buckets.setdefault(key + (k,), []).append(at)
        for (field, nrows, _), members in buckets.items():
            cols, qcol, qpts, spans = [], [], [], []
            for at in members:
                matrix, _, queries = parts[at]
                first = len(cols)
                cols += [c.ptr for c in matrix.columns]
                spans.append((at, len(qcol), len(queries)))
                qcol += [first + c for c, _ in queries]
                qpts += [w for _, p in queries for w in _words(p, self.fq)]
            pts = np.array(qpts, dtype=np.uint64)
            out = np.zeros(len(qcol) * pw, dtype=np.uint64)
            L.check(L.ms_horner_eval(pl.handle, field, self.fq, nrows, (ctypes.c_void_p * len(cols))(*cols), len(cols), (ctypes.c_uint * len(qcol))(*qcol),
                                     pts.ctypes.data, len(qcol), out.ctypes.data))
            for at, q0, nq in spans:
                results[at] = [_from_words(out[pw * i:pw * i + pw]) for i in range(q0, q0 + nq)]
        return results

    def get_ood_evals(self):                      # src/composer.rs:43-86
        base_q = [(c, self._point(o)) for c, o in self.args if c < self.nbase]
        ext_q = [(c - self.nbase, self._point(o)) for c, o in self.args if c >= self.nbase]
        z_n = _q_pow(self.z, self.comp.num_cols(), self.p)
        parts = [(self.base, self.base_field, base_q)] + ([(self.ext, self.fq, ext_q)] if ext_q else []) + \
                [(self.comp, self.fq, [(c, z_n) for c in range(self.comp.num_cols())])]
        res = self._horner(parts)
        bv, ev = iter(res[0]), iter(res[1] if ext_q else [])
        execution = [next(bv) if c < self.nbase else next(ev) for c, _ in self.args]
        composition = res[-1]
        self._ood = (execution, composition)
        return execution, composition

    def _terms(self, coeffs):
        """the DEEP composition's terms (column, point, alpha, P(point)) in the order of src/composer.rs:89-165; columns are numbered
        base | extension | composition-trace"""
        if self._ood is None:
            self.get_ood_evals()
        execution, composition = self._ood
        points, pindex = [], {}

        def pid(p):
            if p not in pindex:
                pindex[p] = len(points)
                points.append(p)
            return pindex[p]
        z_n = _q_pow(self.z, self.comp.num_cols(), self.p)
        tcol, tpoint, talpha, tood = [], [], [], []
        for c in range(self.comp.num_cols()):
            tcol.append(self.nbase + self.next + c); tpoint.append(pid(z_n)); talpha.append(coeffs.composition_trace[c]); tood.append(composition[c])
        for (c, o), alpha, val in zip(self.args, coeffs.execution_trace, execution):
            tcol.append(c); tpoint.append(pid(self._point(o))); talpha.append(alpha); tood.append(val)
        return points, tcol, tpoint, talpha, tood

    def _compose(self, entry, head_args, coeffs, base_cols, ext_cols, comp_cols, out):
        points, tcol, tpoint, talpha, tood = self._terms(coeffs)
        pl, L = self.planner, self.planner.lib
        ext_all = list(ext_cols) + list(comp_cols)
        if self.fq != GOLDILOCKS_FQ3:
            base_list, ext_list = list(base_cols) + ext_all, []              # Fq = Fp: everything is a base column
        else:
            base_list, ext_list = list(base_cols), ext_all
        VP = ctypes.c_void_p
        flat = lambda qs: np.array([w for q in qs for w in _words(q, self.fq)], dtype=np.uint64)
        pts, al, od = flat(points), flat(talpha), flat(tood)
        da, db = flat([coeffs.degree[0]]), flat([coeffs.degree[1]])
        L.check(entry(pl.handle, self.fq, *[a.ctypes.data if isinstance(a, np.ndarray) else a for a in head_args],
                      (VP * max(1, len(base_list)))(*[c.ptr for c in base_list]), len(base_list),
                      (VP * max(1, len(ext_list)))(*[c.ptr for c in ext_list]), len(ext_list),
                      pts.ctypes.data, len(points), (ctypes.c_uint * len(tcol))(*tcol), (ctypes.c_uint * len(tpoint))(*tpoint),
                      al.ctypes.data, od.ctypes.data, len(tcol), da.ctypes.data, db.ctypes.data, out.ptr))
        return out

    def into_deep_poly(self, coeffs):             # src/composer.rs:89-188
        ext_cols = list(self.ext.columns) if self.ext is not None else []
        return self._compose(self.planner.lib.ms_deep_compose, (self.n.bit_length() - 1, None), coeffs, self.base.columns, ext_cols,
                             self.comp.columns, GpuVec(self.planner, self.n, self.fq))

    def into_deep_evaluations(self, coeffs, base_lde, ext_lde, comp_lde, domain_size, first=0, offset=None):
        """`into_deep_poly(coeffs)` followed by `into_bit_reversed_evaluations(lde_domain)` (src/prover.rs:149-152) in one step and
        without the transforms: the composition polynomial's values at rows [first, first + count) of the bit-reversed LDE domain
        (domain_size points, coset offset `offset`, default 7), computed from those rows of the committed LDE matrices -- the same
        field elements (the quotient is a polynomial).  The matrices hold `count` rows: the whole domain on one GPU, or a rank's
        row shard (ministark_amd.distributed.prove_sharded)."""
        if self.base_field == STARK252_FP:
            raise ValueError("into_deep_evaluations: Goldilocks fields")
        count = base_lde.num_rows()
        ext_cols = list(ext_lde.columns) if ext_lde is not None else []
        for m in ([base_lde, comp_lde] + ([ext_lde] if ext_lde is not None else [])):
            if m.num_rows() != count:
                raise ValueError("LDE matrices of different heights")
        off = None
        if offset is not None:
            from .api import _offset_words
            off = _offset_words(GOLDILOCKS_FP, offset)          # (kept alive by head_args until the call has returned)
        return self._compose(self.planner.lib.ms_deep_rows, (domain_size.bit_length() - 1, off, first, count), coeffs, base_lde.columns,
                             ext_cols, comp_lde.columns, GpuVec(self.planner, count, self.fq))
