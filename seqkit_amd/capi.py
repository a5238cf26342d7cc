"""ctypes binding of include/seqkit_hip.h (one-to-one, no logic of its own)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SKH_LIB") or os.path.join(HERE, "libseqkit_hip.so")

MAX_CLASSES = 8
NO_CLASS = 0xFF

# every symbol declared in include/seqkit_hip.h: name -> (restype, argtypes)
u8p, u64p, vp = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.c_void_p


class StatsPartial(C.Structure):
    _fields_ = [("num", C.c_int64), ("len_sum", C.c_int64), ("gap_sum", C.c_int64),
                ("len_min", C.c_int64), ("len_max", C.c_int64)]

    def as_tuple(self):
        return (self.num, self.len_sum, self.gap_sum, self.len_min, self.len_max)


class ParseResult(C.Structure):
    _fields_ = [("nrec", C.c_uint64), ("seq_bytes", C.c_uint64), ("consumed", C.c_uint64), ("lines", C.c_uint64),
                ("error_line", C.c_int64), ("status", C.c_uint32), ("wrapped", C.c_uint32)]


class SeqFormat(C.Structure):
    _fields_ = [("min_len", C.c_int64), ("max_len", C.c_int64), ("line_width", C.c_int32),
                ("print_name", C.c_uint8), ("print_seq", C.c_uint8), ("print_qual", C.c_uint8),
                ("plus_line", C.c_uint8), ("only_id", C.c_uint8), ("mark", C.c_char), ("empty_as_fasta", C.c_uint8)]


class TabColumn(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("nterm", C.c_uint8), ("term", C.c_uint8 * 30)]


class TabFormat(C.Structure):
    _fields_ = [("ncols", C.c_uint32), ("only_id", C.c_uint8), ("print_seq", C.c_uint8), ("print_qual", C.c_uint8),
                ("alphabet", C.c_uint8), ("cols", TabColumn * 16), ("d_letters", C.c_void_p)]


TAB_LENGTH, TAB_GC, TAB_GC_SKEW, TAB_CONTENT = 0, 1, 2, 3
OP_REVCOMP, OP_REVCOMP_INPLACE, OP_COUNT = 1, 2, 3


class Slice(C.Structure):
    _fields_ = [("record", C.c_uint64), ("start", C.c_int64), ("end", C.c_int64), ("minus", C.c_uint32),
                ("reserved", C.c_uint32)]


class SubseqResult(C.Structure):
    _fields_ = [("out_bytes", C.c_uint64), ("status", C.c_uint32), ("reserved", C.c_uint32), ("bad_slice", C.c_uint64)]


class ValidateResult(C.Structure):
    _fields_ = [("record", C.c_uint64), ("offset", C.c_uint64), ("letter", C.c_uint32), ("found", C.c_uint32)]


class ShareInfo(C.Structure):
    _fields_ = [("first_byte", C.c_uint64), ("last_byte", C.c_uint64), ("first_record", C.c_uint64),
                ("nrecords", C.c_uint64), ("head_cut", C.c_uint64), ("tail_cut", C.c_uint64),
                ("chunks", C.c_uint32), ("reserved", C.c_uint32), ("seconds", C.c_double)]


class FormatResult(C.Structure):
    _fields_ = [("out_bytes", C.c_uint64), ("status", C.c_uint32), ("reserved", C.c_uint32)]


FASTA, FASTQ, FASTQ_WRAPPED = 1, 2, 3
PARSE_OK, PARSE_EFORMAT, PARSE_ECAPACITY = 0, 1, 2

SYMBOLS = {
    "skh_last_error": (C.c_char_p, []),
    "skh_version": (C.c_char_p, []),
    "skh_padded_size": (C.c_uint64, [C.c_uint64]),
    "skh_init": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "skh_destroy": (None, [vp]),
    "skh_sync": (C.c_int, [vp]),
    "skh_pair_table": (C.c_int, [C.c_int, u8p]),
    "skh_class_table": (C.c_int, [C.POINTER(C.c_char_p), C.c_uint32, u8p]),
    "skh_revcomp_dev": (C.c_int, [vp, vp, vp, vp, C.c_uint64, C.c_uint64, u8p, C.c_int, vp]),
    "skh_revcomp_fastq_dev": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_uint64, C.c_uint64, u8p, C.c_int, vp]),
    "skh_revcomp_inplace_dev": (C.c_int, [vp, vp, vp, vp, C.c_uint64, C.c_uint64, u8p, C.c_int, vp]),
    "skh_count_dev": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint64, u8p, C.c_uint32, vp, vp]),
    "skh_stats_dev": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint64, u8p, vp, vp]),
    "skh_stats_static_dev": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint64, u8p, vp, vp]),
    "skh_kernel_for": (C.c_int, [vp, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_char_p, C.c_uint64]),
    "skh_parse_workspace_size": (C.c_uint64, [C.c_uint64, C.c_uint64]),
    "skh_parse_wrapped_workspace_size": (C.c_uint64, [C.c_uint64, C.c_uint64, C.c_uint64]),
    "skh_parse_dev": (C.c_int, [vp, vp, C.c_uint64, C.c_int, C.c_int, vp, vp, C.c_uint64, vp, vp, vp, C.c_uint64,
                                vp, C.c_uint64, C.c_uint64, vp, vp]),
    "skh_format_workspace_size": (C.c_uint64, [C.c_uint64]),
    "skh_format_seq_dev": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_uint64, C.POINTER(SeqFormat), vp, C.c_uint64,
                                     vp, C.c_uint64, vp, vp]),
    "skh_format_tab_dev": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_uint64, C.POINTER(vp), C.POINTER(C.c_uint32),
                                     C.POINTER(TabFormat), vp, C.c_uint64, vp, C.c_uint64, vp, vp]),
    "skh_text_count": (C.c_int, [vp, vp, u8p, C.c_uint32, C.c_uint32]),
    "skh_text_format_tab": (C.c_int, [vp, vp, C.POINTER(TabFormat), C.POINTER(vp), u64p]),
    "skh_valid_table": (C.c_int, [C.c_int, u8p]),
    "skh_validate_dev": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint64, u8p, C.c_uint64, vp, vp]),
    "skh_validate": (C.c_int, [vp, vp, C.c_uint64, u8p, C.c_uint64, C.POINTER(ValidateResult)]),
    "skh_text_validate": (C.c_int, [vp, vp, u8p, C.c_uint64, C.POINTER(ValidateResult)]),
    "skh_text_truncate": (C.c_int, [vp, vp, C.c_uint64]),
    "skh_text_name": (C.c_int, [vp, vp, C.c_uint64, u64p, C.POINTER(C.c_uint32)]),
    "skh_quality_table": (C.c_int, [C.c_int, C.c_int, C.c_int, u8p, u8p, C.POINTER(C.c_int)]),
    "skh_text_quality": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64, u8p]),
    "skh_text_validate_quality": (C.c_int, [vp, vp, u8p, C.POINTER(ValidateResult)]),
    "skh_text_map_quality": (C.c_int, [vp, vp, u8p]),
    "skh_subseq_workspace_size": (C.c_uint64, [C.c_uint64]),
    "skh_subseq_dev": (C.c_int, [vp, vp, vp, vp, C.c_uint64, C.c_uint64, vp, C.c_uint64, C.c_int64, C.c_int64, C.c_int,
                                 u8p, vp, vp, C.c_uint64, vp, vp, C.c_uint64, vp, vp]),
    "skh_text_names": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64, u64p, C.POINTER(C.c_uint32)]),
    "skh_text_set_names": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64, u64p, C.POINTER(C.c_uint32)]),
    "skh_text_subseq": (C.c_int, [vp, vp, C.POINTER(Slice), C.c_uint64, C.c_int64, C.c_int64, C.c_int, u8p, u8p,
                                  C.c_uint64, u64p, C.POINTER(C.c_uint32)]),
    "skh_text_alloc": (C.c_int, [vp, C.c_uint64, C.POINTER(vp)]),
    "skh_text_free": (None, [vp]),
    "skh_text_buffer": (vp, [vp]),
    "skh_text_capacity": (C.c_uint64, [vp]),
    "skh_text_parse": (C.c_int, [vp, vp, C.c_uint64, C.c_int, C.c_int, C.POINTER(ParseResult)]),
    "skh_text_parse_at": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.POINTER(ParseResult)]),
    "skh_text_upload": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64]),
    "skh_text_offsets": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64, u64p]),
    "skh_text_sequence": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64, u8p]),
    "skh_text_revcomp": (C.c_int, [vp, vp, u8p, C.c_int, C.c_int]),
    "skh_text_format_seq": (C.c_int, [vp, vp, C.POINTER(SeqFormat), C.POINTER(vp), u64p]),
    "skh_text_format_seq_submit": (C.c_int, [vp, vp, C.POINTER(SeqFormat)]),
    "skh_text_format_tab_submit": (C.c_int, [vp, vp, C.POINTER(TabFormat)]),
    "skh_text_output_wait": (C.c_int, [vp, vp, C.POINTER(vp), u64p]),
    "skh_remove_gaps_workspace_size": (C.c_uint64, [C.c_uint64, C.c_uint64]),
    "skh_remove_gaps_dev": (C.c_int, [vp, vp, vp, vp, C.c_uint64, C.c_uint64, u8p, vp, vp, vp, vp, C.c_uint64, vp, vp]),
    "skh_text_remove_gaps": (C.c_int, [vp, vp, u8p]),
    "skh_letters_dev": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint64, vp, vp]),
    "skh_text_letters": (C.c_int, [vp, vp]),
    "skh_text_stats": (C.c_int, [vp, vp, u8p, C.POINTER(StatsPartial)]),
    "skh_text_ctx": (vp, [vp]),
    "skh_split_by_bytes": (C.c_int, [u64p, C.c_uint64, C.c_int, u64p]),
    "skh_split_record": (C.c_int, [C.c_uint64, C.c_int, u64p]),
    "skh_comm_init_all": (C.c_int, [C.POINTER(vp), C.c_int, C.POINTER(vp)]),
    "skh_comm_unique_id": (C.c_int, [u8p]),
    "skh_comm_init_rank": (C.c_int, [vp, u8p, C.c_int, C.c_int, C.POINTER(vp)]),
    "skh_comm_destroy": (None, [vp]),
    "skh_comm_local_ranks": (C.c_int, [vp]),
    "skh_comm_rccl_version": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "skh_multi_comm_init": (C.c_int, [vp]),
    "skh_stats_allreduce": (C.c_int, [vp, C.POINTER(StatsPartial)]),
    "skh_stats_allreduce_dev": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp)]),
    "skh_split_plane": (C.c_int, [u64p, C.c_uint64, C.c_int, u64p]),
    "skh_multi_init": (C.c_int, [C.POINTER(vp), C.c_int, C.POINTER(vp)]),
    "skh_multi_destroy": (None, [vp]),
    "skh_multi_size": (C.c_int, [vp]),
    "skh_revcomp_multi": (C.c_int, [vp, vp, C.c_uint64, u8p, C.c_int, C.c_int]),
    "skh_count_multi": (C.c_int, [vp, vp, C.c_uint64, u8p, C.c_uint32, u64p]),
    "skh_stats_multi": (C.c_int, [vp, vp, C.c_uint64, u8p, C.POINTER(StatsPartial)]),
    "skh_multi_shares": (C.c_int, [vp, C.POINTER(ShareInfo)]),
    "skh_batch_alloc": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(vp)]),
    "skh_batch_free": (None, [vp]),
    "skh_batch_seq": (vp, [vp]),
    "skh_batch_qual": (vp, [vp]),
    "skh_batch_offsets": (vp, [vp]),
    "skh_batch_max_bytes": (C.c_uint64, [vp]),
    "skh_batch_max_records": (C.c_uint64, [vp]),
    "skh_revcomp": (C.c_int, [vp, vp, C.c_uint64, u8p, C.c_int, C.c_int]),
    "skh_count": (C.c_int, [vp, vp, C.c_uint64, u8p, C.c_uint32, u64p]),
    "skh_stats_init": (None, [C.POINTER(StatsPartial)]),
    "skh_stats_reduce": (C.c_int, [vp, vp, C.c_uint64, u8p, C.POINTER(StatsPartial)]),
    "skh_stats_merge": (None, [C.POINTER(StatsPartial), C.POINTER(StatsPartial)]),
    "skh_revcomp_submit": (C.c_int, [vp, vp, C.c_uint64, u8p, C.c_int, C.c_int]),
    "skh_count_submit": (C.c_int, [vp, vp, C.c_uint64, u8p, C.c_uint32, u64p]),
    "skh_batch_wait": (C.c_int, [vp]),
}


class SkhError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("seqkit_hip error %d: %s" % (code, msg))
        self.code = code


class Lib:
    """The loaded library with typed entry points; `check` raises on a negative return."""

    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise ImportError("%s is missing: build it with `python seqkit_amd/build.py` "
                              "(there is no CPU fallback)" % path)
        self.dll = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(self.dll, name)      # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)

    def check(self, rc):
        if rc != 0:
            raise SkhError(rc, self.skh_last_error().decode())
        return rc

    # small conveniences used by tests/bench (argument marshalling only)
    @staticmethod
    def table(values):
        arr = (C.c_uint8 * 256)(*values)
        return arr

    def pair_table(self, alphabet):
        t = (C.c_uint8 * 256)()
        self.check(self.skh_pair_table(alphabet, t))
        return t

    def class_table(self, lists):
        arr = (C.c_char_p * len(lists))(*[s.encode() if isinstance(s, str) else s for s in lists])
        t = (C.c_uint8 * 256)()
        self.check(self.skh_class_table(arr, len(lists), t))
        return t

    def init(self, device=0):
        ctx = vp()
        self.check(self.skh_init(device, C.byref(ctx)))
        return ctx


_lib = None


def load():
    global _lib
    if _lib is None:
        _lib = Lib()
    return _lib
