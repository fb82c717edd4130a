// Package gpu binds libpqdecode.so (include/pqdecode.h) into hexbee-net/parquet-go with cgo.
//
// Copy this directory to <parquet-go>/gpu, the header to third_party/pqdecode/include and the
// library to third_party/pqdecode/lib (or point the two #cgo lines elsewhere).  Each function
// names the reference code whose loop it replaces.  The file is plain cgo with no build tags; it
// has not been compiled in this tree (the build image carries no Go toolchain), the C side it
// calls is exercised by the C++ and Python hosts' tests.
package gpu

/*
#cgo CFLAGS:  -I${SRCDIR}/../third_party/pqdecode/include
#cgo LDFLAGS: -L${SRCDIR}/../third_party/pqdecode/lib -lpqdecode
#include "pqdecode.h"
*/
import "C"

import (
	"io"
	"runtime"
	"unsafe"

	"github.com/hexbee-net/errors"
)

// Ctx wraps one pq_ctx.  A pq_ctx is single-threaded: keep one per worker goroutine that has
// called runtime.LockOSThread, or guard it with a mutex.
type Ctx struct{ c *C.pq_ctx }

func NewCtx(device int) (*Ctx, error) {
	var c *C.pq_ctx
	if st := C.pq_ctx_create(C.int(device), nil, &c); st != C.PQ_OK {
		return nil, errors.New(C.GoString(C.pq_status_string(st)))
	}
	ctx := &Ctx{c}
	runtime.SetFinalizer(ctx, func(x *Ctx) { C.pq_ctx_destroy(x.c) })
	return ctx, nil
}

// SetOption sets one of the PQ_OPT_* knobs of pqdecode.h (segment length, page limits of the
// segmented decodes, groups of a plan side by side, ...) for the plans this context creates next.
func (x *Ctx) SetOption(option int, value int64) error {
	return statusErr(C.pq_ctx_set_option(x.c, C.int(option), C.int64_t(value)))
}

// statusErr maps the integer status back to the errors the Go decoders return, so callers that
// test errors.Cause(err) == io.EOF keep working.
func statusErr(st C.int) error {
	switch st {
	case C.PQ_OK:
		return nil
	case C.PQ_EOF:
		return io.EOF
	case C.PQ_UNEXPECTED_EOF:
		return io.ErrUnexpectedEOF
	default:
		return errors.New(C.GoString(C.pq_status_string(st)))
	}
}

// DeltaInt64 replaces the loop of Int64DeltaBPDecoder.DecodeValues (types/int64.go:109-121):
// src is everything the bytes.Reader still holds; returns (count, consumed, err).
func (x *Ctx) DeltaInt64(src []byte, dst []int64) (int, int, error) {
	var n C.uint32_t
	var used C.uint64_t
	st := C.pq_delta_decode64(x.c, (*C.uint8_t)(unsafe.Pointer(unsafe.SliceData(src))), C.size_t(len(src)),
		C.uint32_t(len(dst)), (*C.int64_t)(unsafe.Pointer(unsafe.SliceData(dst))), &n, &used)
	return int(n), int(used), statusErr(st)
}

// DictInt32 replaces DictDecoder.DecodeValues (types/dict.go:91-117) for an INT32 dictionary.
// src starts at the bit-width byte, i.e. it is what DictDecoder.Init was handed.
func (x *Ctx) DictInt32(src []byte, dict []int32, dst []int32) (int, error) {
	var n C.uint32_t
	st := C.pq_dict_decode(x.c, (*C.uint8_t)(unsafe.Pointer(unsafe.SliceData(src))), C.size_t(len(src)),
		unsafe.Pointer(unsafe.SliceData(dict)), C.uint32_t(len(dict)), 4, C.uint32_t(len(dst)),
		unsafe.Pointer(unsafe.SliceData(dst)), &n, nil)
	return int(n), statusErr(st)
}

// Levels replaces decodePackedArray (layout/helpers.go:68-90): mode 0 = ConstDecoder, 1 = V1
// length-prefixed stream, 2 = V2 raw stream.  Returns the int32 levels and notNull.
func (x *Ctx) Levels(src []byte, maxLevel uint16, mode int, count int) ([]int32, int, error) {
	lv := make([]int32, count)
	var notNull C.uint32_t
	st := C.pq_levels_decode(x.c, (*C.uint8_t)(unsafe.Pointer(unsafe.SliceData(src))), C.size_t(len(src)),
		C.uint32_t(maxLevel), C.int(mode), C.uint32_t(count), (*C.int32_t)(unsafe.Pointer(unsafe.SliceData(lv))),
		&notNull, nil)
	return lv, int(notNull), statusErr(st)
}

// DecodePages replaces the per-page loop of readPageData (file-reader.go:268-302) for one column
// chunk: descs describes every data page of the chunk inside `block`.
func (x *Ctx) DecodePages(descs []C.pq_page_desc, block, values []byte, levels []int32, dict []byte,
	results []C.pq_page_result) error {
	st := C.pq_decode_pages(x.c, &descs[0], C.size_t(len(descs)),
		unsafe.Pointer(unsafe.SliceData(block)), C.size_t(len(block)),
		unsafe.Pointer(unsafe.SliceData(values)), C.size_t(len(values)),
		unsafe.Pointer(unsafe.SliceData(levels)), C.size_t(4*len(levels)),
		unsafe.Pointer(unsafe.SliceData(dict)), C.size_t(len(dict)), &results[0])
	return statusErr(st)
}

// ByteArrays replaces ByteArrayDeltaLengthDecoder / ByteArrayDeltaDecoder (types/byte-array.go:168-348):
// one call for Init + DecodeValues(dest[:n]).  front selects DELTA_BYTE_ARRAY; room bounds the rebuilt
// values (ignored for DELTA_LENGTH_BYTE_ARRAY, whose values are a slice of src).
func (x *Ctx) ByteArrays(src []byte, n int, front bool, room int) (vals [][]byte, initFailed bool, err error) {
	offsets := make([]uint32, n+1)
	if !front {
		room = len(src)
	}
	data := make([]byte, room+1)
	var cnt C.uint32_t
	var bad C.int
	var st C.int
	sp := (*C.uint8_t)(unsafe.Pointer(unsafe.SliceData(src)))
	op := (*C.uint32_t)(unsafe.Pointer(unsafe.SliceData(offsets)))
	dp := (*C.uint8_t)(unsafe.Pointer(unsafe.SliceData(data)))
	if front {
		st = C.pq_delta_ba_decode(x.c, sp, C.size_t(len(src)), C.uint32_t(n), op, dp, C.size_t(room), &cnt, nil, &bad)
	} else {
		st = C.pq_delta_length_ba_decode(x.c, sp, C.size_t(len(src)), C.uint32_t(n), op, dp, &cnt, nil, &bad)
	}
	vals = make([][]byte, int(cnt))
	for i := range vals {
		vals[i] = data[offsets[i]:offsets[i+1]:offsets[i+1]]
	}
	return vals, bad != 0, statusErr(st)
}

// WalkChunks replaces ChunkReader.readPages (layout/chunk.go:129-217) for every chunk of a row group:
// src holds the chunks' bytes (host memory here; pq_walk_chunks_dev takes device memory).
func (x *Ctx) WalkChunks(src []byte, chunks []C.pq_chunk_desc, infos []C.pq_page_info) ([]C.pq_chunk_result, error) {
	res := make([]C.pq_chunk_result, len(chunks))
	st := C.pq_walk_chunks(x.c, unsafe.Pointer(unsafe.SliceData(src)), C.size_t(len(src)), &chunks[0],
		C.size_t(len(chunks)), &infos[0], C.size_t(len(infos)), &res[0])
	return res, statusErr(st)
}

// Plan is a descriptor table kept on the device (pq_plan_create): made once per chunk layout, run
// as often as buffers with that layout arrive.  RowGroupPlan is the same over the chunks of a whole
// row group, each chunk in buffers of its own (pq_plan_create_sets): the loop over column chunks
// of readRowGroupData (file-reader.go:156-191) becomes one launch.
type Plan struct {
	p *C.pq_plan
	n int
}

// NewRowGroupPlan: descs holds the page descriptors of every chunk back to back, firstDesc[c] is
// the index of chunk c's first descriptor (len(firstDesc) = chunks + 1, as pq_chunks_page_descs
// fills it); offsets in a chunk's descriptors are relative to that chunk's buffers.
func (x *Ctx) NewRowGroupPlan(descs []C.pq_page_desc, firstDesc []uint32) (*Plan, error) {
	var p *C.pq_plan
	st := C.pq_plan_create_sets(x.c, &descs[0], C.size_t(len(descs)),
		(*C.uint32_t)(unsafe.Pointer(unsafe.SliceData(firstDesc))), C.size_t(len(firstDesc)-1), &p)
	if st != C.PQ_OK {
		return nil, statusErr(st)
	}
	return &Plan{p, len(descs)}, nil
}

// Run decodes every chunk: sets[c] names chunk c's device buffers (C memory or device pointers
// only: no Go pointers inside the table).  Asynchronous on the context's stream.
func (pl *Plan) Run(sets []C.pq_buffer_set) error {
	return statusErr(C.pq_plan_run_sets(pl.p, &sets[0], C.size_t(len(sets))))
}

// Results waits for the stream and returns one record per descriptor, in descriptor order.
func (pl *Plan) Results() ([]C.pq_page_result, error) {
	res := make([]C.pq_page_result, pl.n)
	st := C.pq_plan_results(pl.p, &res[0])
	return res, statusErr(st)
}

func (pl *Plan) Close() { C.pq_plan_destroy(pl.p) }

// ColumnEntries replaces the per-row loop of ColumnStore.Get (datastore/column.go:127-177) for a
// chunk whose int32 levels are in host memory: entry k is null when validity bit k is clear (its
// definition level is entryDef[k]), else values[offsets[k]:offsets[k+1]].
func (x *Ctx) ColumnEntries(def, rep []int32, maxD, maxR uint32, repeated bool) (offsets []uint32, validity []byte,
	entryDef []int32, err error) {
	n := len(def)
	offsets = make([]uint32, n+1)
	validity = make([]byte, (n+7)/8+1)
	entryDef = make([]int32, n+1)
	var res C.pq_entries_result
	var rp *C.int32_t
	if repeated {
		rp = (*C.int32_t)(unsafe.Pointer(unsafe.SliceData(rep)))
	}
	rflag := C.int(0)
	if repeated {
		rflag = 1
	}
	st := C.pq_column_entries(x.c, (*C.int32_t)(unsafe.Pointer(unsafe.SliceData(def))), rp, C.uint64_t(n),
		C.uint32_t(maxD), C.uint32_t(maxR), rflag, (*C.uint32_t)(unsafe.Pointer(unsafe.SliceData(offsets))),
		(*C.uint8_t)(unsafe.Pointer(unsafe.SliceData(validity))), (*C.int32_t)(unsafe.Pointer(unsafe.SliceData(entryDef))), &res)
	if st != C.PQ_OK {
		return nil, nil, nil, statusErr(st)
	}
	k := int(res.n_entries)
	return offsets[:k+1], validity[:(k+7)/8], entryDef[:k], statusErr(C.int(res.status))
}
