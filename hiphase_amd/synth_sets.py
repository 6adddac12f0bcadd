"""Synthetic read-bearing block sets straight in the C layout (include/hiphase_gpu.h `hp_synth_reads_*`,
hiphase_amd/csrc/hp_synth_reads.cpp): generated, solved and compared without any Python marshalling in between.
`dll` is the library that generates (the product library by default; the test oracle carries the same generator)."""
import ctypes as C

from . import _ffi


def default_spec(dll=None, hifi=False, deep60=False, **kw):
    """The bench workload's spec (uniform 0.5 % edit noise); hifi=True: HiFi-shaped errors (hp_synth_reads_hifi: per-read rate
    lognormal around 0.2 %, half of the errors homopolymer indels); deep60=True: BASELINE.json configs[4]'s shape (hp_synth_reads_deep60:
    60x, 15 % wrong-haplotype cells, every tandem-repeat het multi-allelic)."""
    dll = dll or _ffi.lib()
    s = _ffi.SynthReadsSpec()
    (dll.hp_synth_reads_deep60 if deep60 else dll.hp_synth_reads_hifi if hifi else dll.hp_synth_reads_defaults)(C.byref(s))
    for k, v in kw.items():
        if not hasattr(s, k):
            raise AttributeError(k)
        setattr(s, k, v)
    return s


class SynthSet:
    """One generated block set: `.inputs` (hp_block_input array), `.n`, `.info`; `.outputs()` makes caller-side result buffers."""

    def __init__(self, spec, dll=None):
        self.dll = dll or _ffi.lib()
        st = C.c_int(0)
        self.h = self.dll.hp_synth_reads_create(C.byref(spec), C.byref(st))
        if not self.h:
            raise _ffi.HpError(st.value, "hp_synth_reads_create")
        n = C.c_size_t(0)
        self.inputs = self.dll.hp_synth_reads_inputs(self.h, C.byref(n))
        self.n = n.value
        info = (C.c_uint64 * 8)()
        self.dll.hp_synth_reads_info(self.h, info)
        self.info = dict(zip(("blocks", "hets", "records", "read_bases", "qnames", "input_bytes", "max_block_hets"), list(info)))

    def outputs(self):
        return Outputs(self.dll, self.inputs, self.n)

    def relocate_pinned(self):
        """Moves the records' bases into memory from hp_host_alloc (pinned, device-readable: the library then reads them in
        place instead of staging them - what a loader that decodes BAM records into such an arena gets). The allocator always
        comes from the PRODUCT library, whichever library generated the set."""
        prod = _ffi.lib()
        rc = self.dll.hp_synth_reads_relocate(self.h, C.cast(prod.hp_host_alloc, C.c_void_p), C.cast(prod.hp_host_free, C.c_void_p))
        if rc != 0:
            raise _ffi.HpError(rc, "hp_synth_reads_relocate")
        return self

    def truth(self, b):
        p = self.dll.hp_synth_reads_truth(self.h, b)
        return [p[i] for i in range(self.inputs[b].n_hets)]

    def close(self):
        if self.h:
            self.dll.hp_synth_reads_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class Outputs:
    """caller-side result buffers for `n` blocks (hp_outputs_*)"""

    def __init__(self, dll, inputs, n):
        self.dll, self.inputs, self.n = dll, inputs, n
        self.h = dll.hp_outputs_create(inputs, n)
        self.arr = dll.hp_outputs_array(self.h)

    def equal(self, other, b):
        return bool(self.dll.hp_block_output_equal(C.byref(self.inputs[b]), C.byref(self.arr[b]), C.byref(other.arr[b])))

    def poison(self, fill):
        """every array and scalar result filled with the byte `fill` (pointers and capacities kept): two sets poisoned with
        different bytes can only compare equal in fields a solve really wrote"""
        self.dll.hp_outputs_poison(self.h, fill)
        return self

    def close(self):
        if self.h:
            self.dll.hp_outputs_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class Capture:
    """A `.hpbr` read-bearing capture loaded back (hp_hpbr_*): `.inputs`, `.n`, `.params`, `.expected` (status INT32_MIN = none)."""

    def __init__(self, path, dll=None):
        self.dll = dll or _ffi.lib()
        st = C.c_int(0)
        self.h = self.dll.hp_hpbr_open(str(path).encode(), C.byref(st))
        if not self.h:
            raise _ffi.HpError(st.value, self.dll.hp_hpbr_last_error().decode())
        n = C.c_size_t(0)
        self.inputs = self.dll.hp_hpbr_inputs(self.h, C.byref(n))
        self.n = n.value
        self.params = self.dll.hp_hpbr_params(self.h)
        self.expected = self.dll.hp_hpbr_expected(self.h)
        self.info = {"blocks": self.n, "hets": sum(self.inputs[b].n_hets for b in range(self.n)),
                     "records": sum(self.inputs[b].n_records for b in range(self.n)),
                     "read_bases": sum(self.inputs[b].records[r].read_len for b in range(self.n) for r in range(self.inputs[b].n_records)),
                     "max_block_hets": max([self.inputs[b].n_hets for b in range(self.n)] or [0])}

    def outputs(self):
        return Outputs(self.dll, self.inputs, self.n)

    def close(self):
        if self.h:
            self.dll.hp_hpbr_close(self.h)
            self.h = None

    def __del__(self):
        self.close()
