"""Helpers for the GPU parity tests: pack many independent cases into one
pq_decode_pages call (through the C ABI) and slice the outputs back per case."""
import numpy as np

from parquet_go_amd import capi

CANARY = 0xA5


class Case:
    """One descriptor's worth of work plus where its outputs landed."""

    def __init__(self, kind, src, n, **kw):
        self.kind = kind
        self.src = np.ascontiguousarray(np.frombuffer(bytes(bytearray(src)), dtype=np.uint8)
                                        if not isinstance(src, np.ndarray) else src, dtype=np.uint8)
        self.n = int(n)
        self.flags = kw.get("flags", 0)
        self.bit_width = kw.get("bit_width", 0)
        self.max_rep = kw.get("max_rep", 0)
        self.max_def = kw.get("max_def", 0)
        self.rep_len = kw.get("rep_len", 0)
        self.def_len = kw.get("def_len", 0)
        self.value_kind = kw.get("value_kind", 0)
        self.dictionary = kw.get("dictionary", None)
        self.plain_dtype = np.dtype(kw["plain_dtype"]) if "plain_dtype" in kw else None  # PLAIN values
        self.src_len = kw.get("src_len", None)  # override (e.g. truncate without copying)
        # byte-array kinds: lengths the offsets area has room for, bytes of room for the values
        self.offs_cap = kw.get("offs_cap", self.n)
        self.data_cap = kw.get("data_cap", None)

    def byte_array_kind(self):
        k = self.value_kind if self.kind in (capi.KIND_PAGE_V1, capi.KIND_PAGE_V2) else self.kind
        return k if k in (capi.KIND_DELTA_LENGTH_BA, capi.KIND_DELTA_BA) else 0

    def elem_size(self):
        k = self.value_kind if self.kind in (capi.KIND_PAGE_V1, capi.KIND_PAGE_V2) else self.kind
        if k == capi.KIND_DICT:
            return self.dictionary.dtype.itemsize
        if k == capi.KIND_PLAIN:
            return self.plain_dtype.itemsize
        return {capi.KIND_HYBRID: 4, capi.KIND_DELTA32: 4, capi.KIND_DELTA64: 8, capi.KIND_BOOL_PLAIN: 4,
                capi.KIND_BOOL_RLE: 4}.get(k, 0)

    def value_dtype(self):
        k = self.value_kind if self.kind in (capi.KIND_PAGE_V1, capi.KIND_PAGE_V2) else self.kind
        if k == capi.KIND_DICT:
            return self.dictionary.dtype
        if k == capi.KIND_PLAIN:
            return self.plain_dtype
        return {capi.KIND_HYBRID: np.int32, capi.KIND_DELTA32: np.int32, capi.KIND_DELTA64: np.int64,
                capi.KIND_BOOL_PLAIN: np.int32, capi.KIND_BOOL_RLE: np.int32}.get(k, np.uint8)


def run_cases(ctx, cases, seed=0):
    """Lay the cases out in shared buffers (random gaps, varying alignment, canary
    bytes between outputs), decode them in ONE call, attach results to each case."""
    rng = np.random.default_rng(seed)
    descs = np.zeros(len(cases), dtype=capi.DESC_DTYPE)
    src_parts, src_pos = [], 0
    dict_parts, dict_pos = [], 0
    val_pos, lvl_pos = 0, 0
    for i, c in enumerate(cases):
        gap = int(rng.integers(0, 24))
        src_parts.append(rng.integers(0, 256, size=gap, dtype=np.uint8))
        src_pos += gap
        d = descs[i]
        d["kind"], d["flags"] = c.kind, c.flags
        d["src_off"] = src_pos
        d["src_len"] = c.src.size if c.src_len is None else c.src_len
        src_parts.append(c.src)
        src_pos += c.src.size
        d["n_values"], d["bit_width"] = c.n, c.bit_width
        d["max_rep"], d["max_def"] = c.max_rep, c.max_def
        d["rep_len"], d["def_len"] = c.rep_len, c.def_len
        d["value_kind"] = c.value_kind
        if c.plain_dtype is not None:
            d["elem_size"] = c.plain_dtype.itemsize
        es = c.elem_size()
        if es:
            val_pos += es * int(rng.integers(0, 5))       # vary 16-byte alignment of outputs
            val_pos = (val_pos + es - 1) // es * es
            d["dst_off"] = val_pos
            c.val_off = val_pos
            val_pos += es * c.n + 8
        if c.byte_array_kind():
            front = c.byte_array_kind() == capi.KIND_DELTA_BA
            c.room = int(c.data_cap) if front else int(d["src_len"])
            val_pos += int(rng.integers(0, 9))
            d["dst_off"] = c.val_off = val_pos
            d["dst_cap"] = c.room if front else 0
            val_pos += c.room + 8
            val_pos = (val_pos + 3) // 4 * 4 + 4 * int(rng.integers(0, 4))
            d["dict_off"] = c.area_off = val_pos
            d["dict_len"] = c.offs_cap
            c.area_words = (2 if front else 1) * c.offs_cap + 1
            val_pos += 4 * c.area_words + 8
        if c.kind in (capi.KIND_LEVELS, capi.KIND_PAGE_V1, capi.KIND_PAGE_V2):
            lvl_pos += 4 * int(rng.integers(0, 5))
            d["def_off"] = lvl_pos
            c.def_off = lvl_pos
            lvl_pos += 4 * c.n + 4
            if c.kind != capi.KIND_LEVELS:
                lvl_pos += 4 * int(rng.integers(0, 5))
                d["rep_off"] = lvl_pos
                c.rep_off = lvl_pos
                lvl_pos += 4 * c.n + 4
        if c.dictionary is not None:
            dd = np.ascontiguousarray(c.dictionary)
            pad = (-dict_pos) % 8
            if pad:
                dict_parts.append(np.zeros(pad, np.uint8))
                dict_pos += pad
            d["dict_off"] = dict_pos
            d["dict_len"] = dd.size
            d["elem_size"] = dd.dtype.itemsize
            dict_parts.append(dd.view(np.uint8).reshape(-1))
            dict_pos += dd.nbytes
    src = np.concatenate(src_parts) if src_parts else np.zeros(0, np.uint8)
    dictionary = np.concatenate(dict_parts) if dict_parts else None
    val_bytes = (val_pos + 15) // 8 * 8
    lvl_bytes = (lvl_pos + 7) // 4 * 4
    results, values, levels = _decode_with_canary(ctx, descs, src, val_bytes, lvl_bytes, dictionary)
    for i, c in enumerate(cases):
        c.result = results[i]
        es = c.elem_size()
        if es:
            c.values = values[c.val_off: c.val_off + es * c.n].view(c.value_dtype())
            c.guard = values[c.val_off + es * c.n: c.val_off + es * c.n + 8]
        if c.byte_array_kind():
            c.ba_data = values[c.val_off: c.val_off + c.room]
            c.guard = values[c.val_off + c.room: c.val_off + c.room + 8]
            c.ba_offsets = values[c.area_off: c.area_off + 4 * (c.offs_cap + 1)].view(np.uint32)
            c.area_guard = values[c.area_off + 4 * c.area_words: c.area_off + 4 * c.area_words + 8]
        if c.kind in (capi.KIND_LEVELS, capi.KIND_PAGE_V1, capi.KIND_PAGE_V2):
            c.def_levels = levels[c.def_off // 4: c.def_off // 4 + c.n]
            c.def_guard = levels[c.def_off // 4 + c.n]
            if c.kind != capi.KIND_LEVELS:
                c.rep_levels = levels[c.rep_off // 4: c.rep_off // 4 + c.n]
                c.rep_guard = levels[c.rep_off // 4 + c.n]
    return cases


def _decode_with_canary(ctx, descs, src, val_bytes, lvl_bytes, dictionary):
    """pq_decode_pages copies its output staging back wholesale, so pre-filling the
    host arrays cannot act as a canary; instead run on device buffers we fill."""
    import ctypes as C

    import torch

    dev = torch.device("cuda:0")
    t_src = torch.from_numpy(np.concatenate([src, np.zeros(16, np.uint8)])).to(dev)
    t_val = torch.full((max(val_bytes, 8),), CANARY, dtype=torch.uint8, device=dev)
    t_lvl = torch.full((max(lvl_bytes, 4),), CANARY, dtype=torch.uint8, device=dev)
    t_dict = (torch.from_numpy(dictionary).to(dev) if dictionary is not None and dictionary.size
              else torch.zeros(8, dtype=torch.uint8, device=dev))
    torch.cuda.synchronize()
    plan = ctx.plan(descs)
    plan.run(t_src.data_ptr(), src.size, t_val.data_ptr(), val_bytes, t_lvl.data_ptr(), lvl_bytes,
             t_dict.data_ptr(), 0 if dictionary is None else dictionary.size)
    results = plan.results()
    plan.close()
    values = t_val.cpu().numpy()
    levels = t_lvl.cpu().numpy().view(np.int32)
    return results, values, levels


def guard_ok(arr):
    a = np.atleast_1d(np.asarray(arr)).view(np.uint8)
    return bool((a == CANARY).all())
