"""Host-side operator layer over the C ABI (include/lens_amd.h).

torch is used only for device memory, streams and the reference's own RNG call; every numeric
step of the hot path runs in the HIP kernels of liblens_amd.so.  Nothing here falls back to the
CPU: tensors must live on a ROCm device and the extension must be built.
"""
import ctypes

import numpy as np
import torch

from . import _lib

INT64_MAX = (1 << 63) - 1


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(t, name):
    if not t.is_cuda:
        raise _lib.LensAmdError(f"{name} must be a device (ROCm) tensor; this build has no CPU path")


# --------------------------------------------------------------------------------------------
# a1: events -> frames                                   (lens/collect_data.py:193-199)
# --------------------------------------------------------------------------------------------
def accumulate_events(xy, frame_off, roi, check=True):
    """xy int16 [n_events,2] (x,y 1-based), frame_off int64 [n_frames+1] -> uint8 [n_frames,roi,roi].
    Raises IndexError where torch indexing in the reference would (check=True synchronises to read the
    flag; check=False returns (frames, first_bad_tensor) without synchronising)."""
    _need_cuda(xy, "xy"); _need_cuda(frame_off, "frame_off")
    assert xy.dtype == torch.int16 and xy.dim() == 2 and xy.shape[1] == 2 and xy.is_contiguous()
    assert frame_off.dtype == torch.int64 and frame_off.is_contiguous()
    n_frames = frame_off.numel() - 1
    frames = torch.empty((n_frames, roi, roi), dtype=torch.uint8, device=xy.device)
    bad = torch.full((1,), INT64_MAX, dtype=torch.int64, device=xy.device)
    st = _lib.lib().lens_accumulate_events(_ptr(xy), _ptr(frame_off), n_frames, roi, _ptr(frames), _ptr(bad), _stream())
    _lib.check(st, "lens_accumulate_events")
    if not check:
        return frames, bad
    b = int(bad.item())
    if b != INT64_MAX:
        x, y = (int(v) for v in xy[b].tolist())
        raise IndexError(f"event {b} (x={x}, y={y}) is out of bounds for a {roi}x{roi} frame")
    return frames


# --------------------------------------------------------------------------------------------
# a2/a3 host pieces: centre-pixel index and threshold codes
# --------------------------------------------------------------------------------------------
def centre_pixel_index(roi_dim, dims):
    """Flat pixel index read by input neuron n = r*dims + c: the one-hot Conv2d(k, stride k) of
    lens/run_model.py:130-137 with k = roi_dim // dims and hot tap (k//2 - 1, k//2 - 1)
    (a negative tap index, k == 1, wraps like Python indexing)."""
    k = roi_dim // dims
    c0 = k // 2 - 1
    if c0 < 0:
        c0 += k
    r = np.arange(dims) * k + c0
    return (r[:, None] * roi_dim + r[None, :]).reshape(-1).astype(np.int32)


def rand_tensor(T, HW):
    """The uniform tensor of lens/src/dataset.py:120-121 (`torch.manual_seed(50); torch.rand(T, H*W)`),
    drawn from a private CPU generator so the caller's global RNG state is left alone."""
    g = torch.Generator(device="cpu")
    g.manual_seed(50)
    return torch.rand(T, HW, generator=g)


def build_threshold_codes(T, roi_dim, sel):
    """uint8 [T,N] codes with  (U[t, sel[n]] < px/255)  <=>  (px >= code[t,n])."""
    U = rand_tensor(T, roi_dim * roi_dim)
    u_sel = U[:, torch.as_tensor(np.asarray(sel), dtype=torch.long)].contiguous()
    codes = torch.empty(u_sel.shape, dtype=torch.uint8)
    st = _lib.lib().lens_build_threshold_codes(_ptr(u_sel), u_sel.shape[0], u_sel.shape[1], _ptr(codes))
    _lib.check(st, "lens_build_threshold_codes")
    return codes


# --------------------------------------------------------------------------------------------
# the SNN forward
# --------------------------------------------------------------------------------------------
class LensEngine:
    """Device-resident LENS network: weights, threshold codes, membrane state and workspaces.

    w1: feature_layer.w.weight [F,N];  w2: output_layer.w.weight [P,F] (or a row range of it for a
    database shard).  State (`v_feat`, `v_out`) persists across `forward_counts(carry=True)` calls,
    which is the reference's behaviour (it never resets its IAF layers, lens/run_model.py:234-242).
    """

    def __init__(self, w1, w2, roi_dim, dims, timebin, device="cuda", max_batch=4096):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.LensAmdError("LensEngine needs a ROCm device; this build has no CPU path")
        _lib.lib()
        w1 = torch.as_tensor(w1, dtype=torch.float32)
        w2 = torch.as_tensor(w2, dtype=torch.float32)
        self.F, self.N = w1.shape
        self.P = w2.shape[0]
        assert w2.shape[1] == self.F and self.N == dims * dims
        self.roi_dim, self.dims, self.T = int(roi_dim), int(dims), int(timebin)
        self.HW = self.roi_dim * self.roi_dim
        self.device = dev
        self.max_batch = int(max_batch)
        sel = centre_pixel_index(self.roi_dim, self.dims)
        self.sel = torch.from_numpy(sel).to(dev)
        codes = build_threshold_codes(self.T, self.roi_dim, sel)
        self.codes = codes.to(dev)
        # input spikes per (input neuron, pixel value): the feature kernel's load-balancing hint
        steps = torch.empty((self.N, 256), dtype=torch.int16)
        _lib.check(_lib.lib().lens_build_active_steps(_ptr(codes), self.T, self.N, _ptr(steps)), "lens_build_active_steps")
        self.active_steps = steps.to(dev)
        # transposed, padded weight layouts the kernels read
        self.w1t = w1.t().contiguous().to(dev)                          # [N][F]
        self.ldw2 = (self.P + 255) // 256 * 256
        w2t = torch.zeros((self.F, self.ldw2), dtype=torch.float32, device=dev)
        w2t[:, :self.P] = w2.t().to(dev)
        self.w2t = w2t
        self.cap = int(_lib.lib().lens_event_capacity_words(self.T, self.F))
        # the spike-stream workspace is sized for the worst case (every feature spikes every step):
        # keep it below ~4 GB by limiting the queries processed per launch
        self.max_batch = max(1, min(self.max_batch, (4 << 30) // (4 * self.cap)))
        self.n_streams = 1
        self.reset_state()
        self._ws_q = 0
        # carried state: the two-pass feature layer (input currents of all queries in parallel, then the integrate-and-
        # fire walk with one wavefront per 64 features) where its tables fit; else the one-wavefront kernel
        self.ldc = (self.F + 63) // 64 * 64
        self._cur = self._fire_scratch = None
        nw = self.ldc // 64                                         # LDS tables of lens_feature_fire (its launch check)
        tables = 2 * ((self.T * nw * 12 + 7) // 8 * 8) + 4 * ((2 * self.T + 1) + (self.cap // 32 + 31) // 32) * 4
        self._two_pass = self.F <= 768 and self.T < (1 << 22) and tables <= 159 * 1024

    # ---- state -------------------------------------------------------------------------
    def reset_state(self, n_streams=1):
        """Zero the membrane potentials (sinabs' reset_states()); n_streams independent carried runs."""
        self.n_streams = int(n_streams)
        self.v_feat = torch.zeros((self.n_streams, self.F), dtype=torch.float32, device=self.device)
        self.v_out = torch.zeros((self.n_streams, self.P), dtype=torch.float32, device=self.device)

    def event_scale(self, carry=False):
        """The `ev_scale` of the spike streams for this mode (the output kernel's tile width depends on it:
        with carried state a tile is walked by one wavefront and narrower tiles keep more of them resident)."""
        return int(_lib.lib().lens_event_scale(self.F, self.P, int(bool(carry)), self.n_streams if carry else 0))

    def tile_places(self, carry=False):
        """Places per LDS weight tile of the output kernel in this mode (0: weights read through L2)."""
        return int(_lib.lib().lens_output_tile_places(self.F, self.P, int(bool(carry)), self.n_streams if carry else 0))

    def layout(self, carry=False):
        """lens_output_layout for this mode: 0 batch, 1 lone wavefronts on LDS tiles, 2 lone wavefronts through L2,
        3 wide LDS tiles shared by 16 carried streams."""
        return int(_lib.lib().lens_output_layout(self.F, self.P, int(bool(carry)), self.n_streams if carry else 0))

    def tail_places(self, carry=False):
        """Places at the end of the database that the shared-tile layout runs as a second launch of lone wavefronts on
        128-place tiles (lens_output_tail_places; 0: none)."""
        return int(_lib.lib().lens_output_tail_places(self.F, self.P, int(bool(carry)), self.n_streams if carry else 0))

    def reads_through_l2(self, carry=False):
        """True when the output layer reads the database through L2 instead of LDS tiles in this mode: no tile fits, or
        carried state with more (tile, stream) wavefronts than LDS-resident tiles (lens_output_reads_through_l2)."""
        return bool(_lib.lib().lens_output_reads_through_l2(self.F, self.P, int(bool(carry)), self.n_streams if carry else 0))

    @property
    def ev_scale(self):
        return self.event_scale(False)

    TWO_PASS_MAX_BYTES = 1 << 30     # largest currents tensor [Q][T][ldc] the carried path allocates

    def _carried_two_pass(self, Q):
        return self._two_pass and Q * self.T * self.ldc * 4 <= self.TWO_PASS_MAX_BYTES

    def _carried_buffers(self, Q):
        if self._cur is None or self._cur.shape[0] < Q:
            # lens_feature_fire prefetches 32 rows ahead: that much slack behind the last query (one more query of rows)
            t4 = (self.T + 3) // 4 * 4                       # layout [q][t / 4][ldc][4]: a query takes T4 rows of ldc floats
            self._cur_store = torch.empty(((Q + 1) * t4 + 32, self.ldc), dtype=torch.float32, device=self.device)
            self._cur = self._cur_store[:Q * t4].view(Q, t4, self.ldc)
        need = int(_lib.lib().lens_feature_fire_scratch_bytes(self.T, self.F, self.n_streams))
        if self._fire_scratch is None or self._fire_scratch.numel() < need:
            self._fire_scratch = torch.empty((need,), dtype=torch.uint8, device=self.device)
        return self._cur, self._fire_scratch

    def _feature_currents(self, frames, cur):
        st = _lib.lib().lens_feature_currents(
            _ptr(frames), frames.shape[0], self.HW, _ptr(self.sel), self.N, _ptr(self.codes), self.T,
            _ptr(self.w1t), self.F, self.F, _ptr(cur), self.ldc, _stream())
        _lib.check(st, "lens_feature_currents")

    def _feature_fire(self, cur, Q, n_streams, events, nwords, scratch, scale):
        st = _lib.lib().lens_feature_fire(
            _ptr(cur), self.ldc, Q, self.T, self.F, _ptr(self.v_feat), n_streams,
            _ptr(events), self.cap, _ptr(nwords), scale, _ptr(scratch), _stream())
        _lib.check(st, "lens_feature_fire")

    def _workspace(self, Q):
        if Q > self._ws_q:
            self.events = torch.empty((Q, self.cap), dtype=torch.int32, device=self.device)
            self.nwords = torch.empty((Q,), dtype=torch.int32, device=self.device)
            self._ws_q = Q

    # ---- kernels -----------------------------------------------------------------------
    def feature_events(self, frames, carry=False, want_spikes=False):
        """frames uint8 [Q,H,W] (device) -> fills self.events/self.nwords; optionally returns the dense
        feature spikes float32 [Q,T,F]."""
        _need_cuda(frames, "frames")
        assert frames.dtype == torch.uint8 and frames.is_contiguous()
        Q = frames.shape[0]
        assert frames[0].numel() == self.HW, "frame size does not match roi_dim"
        self._workspace(Q)
        if carry and not want_spikes and self._carried_two_pass(Q):
            cur, scratch = self._carried_buffers(Q)
            self._feature_currents(frames, cur)
            self._feature_fire(cur, Q, self.n_streams, self.events, self.nwords, scratch, self.event_scale(True))
            return None
        spk = torch.empty((Q, self.T, self.F), dtype=torch.float32, device=self.device) if want_spikes else None
        n_streams = self.n_streams if carry else 0
        st = _lib.lib().lens_feature_events(
            _ptr(frames), Q, self.HW, _ptr(self.sel), self.N, _ptr(self.codes), self.T,
            _ptr(self.w1t), self.F, self.F, _ptr(self.v_feat), n_streams, int(carry),
            _ptr(self.events), self.cap, _ptr(self.nwords), self.event_scale(carry), _ptr(spk), _ptr(self.active_steps), _stream())
        _lib.check(st, "lens_feature_events")
        return spk

    def output_counts(self, Q, carry=False, n_groups=0, out=None):
        """Consumes self.events of the last feature_events call -> spike counts float32 [Q,P]."""
        if out is None:
            out = torch.empty((Q, self.P), dtype=torch.float32, device=self.device)
        assert out.shape == (Q, self.P) and out.is_contiguous()
        groups = self.n_streams if carry else n_groups
        st = _lib.lib().lens_output_counts(
            _ptr(self.events), self.cap, _ptr(self.nwords), Q, _ptr(self.w2t), self.F, self.P, self.ldw2,
            _ptr(self.v_out), self.P, groups, int(carry), _ptr(out), self.P, _stream())
        _lib.check(st, "lens_output_counts")
        return out

    def forward_counts_dense(self, frames):
        """Same result as forward_counts(carry=False), computed with the dense f32-MFMA kernel (comparison path)."""
        Q = frames.shape[0]
        spk = self.feature_events(frames, carry=False, want_spikes=True)
        out = torch.empty((Q, self.P), dtype=torch.float32, device=self.device)
        st = _lib.lib().lens_output_counts_dense(_ptr(spk), Q, self.T, self.F, self.F, _ptr(self.w2t), self.P, self.ldw2,
                                                 _ptr(out), self.P, _stream())
        _lib.check(st, "lens_output_counts_dense")
        return out

    def forward_counts(self, frames, carry=False, n_groups=0):
        """The query loop of LENS.evaluate (lens/run_model.py:229-246) for a batch of frames:
        uint8 [Q,H,W] -> similarity-matrix rows float32 [Q,P].  carry=True integrates the queries
        in order with the membrane state carried across them (and across calls); the split into carried streams is
        the engine's `n_streams` (reset_state), `n_groups` applies to the batch mode only."""
        Q = frames.shape[0]
        out = torch.empty((Q, self.P), dtype=torch.float32, device=self.device)
        step = self.max_batch
        if carry and self.n_streams > 1 and Q > step:
            step = Q   # several carried streams split the whole batch; do not re-split it
        for q0 in range(0, Q, step):
            fr = frames[q0:q0 + step]
            if carry and self.n_streams == 1 and fr.shape[0] > self.CARRY_CHUNK:
                self._carried_stream_pipelined(fr, out[q0:q0 + step])
                continue
            self.feature_events(fr, carry=carry)
            self.output_counts(fr.shape[0], carry=carry, n_groups=n_groups, out=out[q0:q0 + step])
        return out

    CARRY_CHUNK = 16    # queries per launch of the pipelined carried stream (8: +4 %, 32: +2 %, 64: +10 % per 256 queries)
    CARRY_FIRST = 4     # ... of the first launch, doubling from there up to CARRY_CHUNK (the output layer idles until the first
                        # chunk's streams exist; 100 000 places: 5.87 -> 5.65 ms per 256 queries, 1000 places 4.86 -> 4.85)

    def _carried_stream_pipelined(self, frames, out):
        """One carried stream (the reference's semantics): both layers are sequential in the query index, each a
        single wavefront per tile, but the output layer of a chunk of queries needs only that chunk's spike streams.
        The feature layer therefore runs one chunk of queries ahead on a second HIP stream (same kernels, same order
        per layer, identical results); the state of each layer is carried from launch to launch in v_feat / v_out."""
        _need_cuda(frames, "frames")
        assert frames.dtype == torch.uint8 and frames.is_contiguous()
        Q = frames.shape[0]
        assert frames[0].numel() == self.HW, "frame size does not match roi_dim"
        self._workspace(Q)
        L = _lib.lib()
        main = torch.cuda.current_stream(self.device)
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
            self._side_done = [torch.cuda.Event(), torch.cuda.Event()]   # chunk k uses event k & 1
        side = self._side
        scale = self.event_scale(True)
        # The side stream uses the spike-stream workspace, v_feat and the caller's frames without telling the caching
        # allocator (no record_stream).  That is sound only because EVERY side-stream launch is joined to the main
        # stream before this method returns -- also when a launch fails half way (the finally below): after the join,
        # frees and re-allocations ordered on the main stream (reset_state, a larger workspace) come after the side
        # stream's work.  An event of the pair is re-recorded two chunks later, after the main stream's wait on it
        # was enqueued (a wait refers to the record that preceded it).
        two_pass = self._carried_two_pass(Q)
        if two_pass:
            cur, scratch = self._carried_buffers(Q)      # allocated on the main stream, before the side stream waits for it
        side.wait_stream(main)          # frames are ready, and earlier readers of the workspace have finished
        try:
            if two_pass:                # the input currents of the whole call, in parallel, ahead of the sequential walk
                with torch.cuda.stream(side):
                    self._feature_currents(frames, cur)
            starts, size = [], min(self.CARRY_FIRST, self.CARRY_CHUNK)
            while not starts or starts[-1][1] < Q:                    # chunk sizes double from CARRY_FIRST up to CARRY_CHUNK
                c0 = starts[-1][1] if starts else 0
                starts.append((c0, min(Q, c0 + size)))
                size = min(2 * size, self.CARRY_CHUNK)
            for k, (c0, c1) in enumerate(starts):
                fr, ev, nw = frames[c0:c1], self.events[c0:c1], self.nwords[c0:c1]
                done = self._side_done[k & 1]
                with torch.cuda.stream(side):
                    if two_pass:
                        self._feature_fire(cur[c0:c1], c1 - c0, 1, ev, nw, scratch, scale)
                    else:
                        st = L.lens_feature_events(
                            _ptr(fr), c1 - c0, self.HW, _ptr(self.sel), self.N, _ptr(self.codes), self.T,
                            _ptr(self.w1t), self.F, self.F, _ptr(self.v_feat), 1, 1,
                            _ptr(ev), self.cap, _ptr(nw), scale, None, None, _stream())
                        _lib.check(st, "lens_feature_events")
                    done.record(side)
                main.wait_event(done)
                o = out[c0:c1]
                st = L.lens_output_counts(
                    _ptr(ev), self.cap, _ptr(nw), c1 - c0, _ptr(self.w2t), self.F, self.P, self.ldw2,
                    _ptr(self.v_out), self.P, 1, 1, _ptr(o), self.P, _stream())
                _lib.check(st, "lens_output_counts")
        finally:
            main.wait_stream(side)      # join: nothing of this call is left running on the side stream unseen by main
            side.wait_stream(main)      # the next call's feature launches must not overtake these output launches

    def forward_counts_static(self, frames, out, carry=False):
        """forward_counts through lens_forward_counts_static: the spike-stream workspace is the library's static
        device array, and this call allocates nothing (the caller passes `out`, float32 [Q,P] on the device).
        For callers that cannot allocate; the batch must fit the array (LensAmdError otherwise)."""
        _need_cuda(frames, "frames"); _need_cuda(out, "out")
        assert frames.dtype == torch.uint8 and frames.is_contiguous()
        Q = frames.shape[0]
        assert Q == 0 or frames[0].numel() == self.HW, "frame size does not match roi_dim"
        assert out.dtype == torch.float32 and out.shape == (Q, self.P) and out.is_contiguous()
        st = _lib.lib().lens_forward_counts_static(
            _ptr(frames), Q, self.HW, _ptr(self.sel), self.N, _ptr(self.codes), self.T,
            _ptr(self.w1t), self.F, self.F, _ptr(self.w2t), self.P, self.ldw2,
            _ptr(self.v_feat), _ptr(self.v_out), self.P, self.n_streams if carry else 0, int(carry),
            _ptr(out), self.P, _stream())
        _lib.check(st, "lens_forward_counts_static")
        return out

    # ---- matching ----------------------------------------------------------------------
    def seqmatch(self, counts, L):
        """lens/run_model.py:248-252, query-major: S[i][j] float32 [Q-L+1, P-L+1]."""
        return seqmatch(counts, L)

    def seqmatch_topk(self, counts, L, K, n_places=None, place_offset=0, out=None):
        return seqmatch_topk(counts, L, K, n_places=n_places, place_offset=place_offset, out=out)


def seqmatch(counts, L):
    _need_cuda(counts, "counts")
    assert counts.dtype == torch.float32 and counts.is_contiguous()
    Q, P = counts.shape
    S = torch.empty((Q - L + 1, P - L + 1), dtype=torch.float32, device=counts.device)
    st = _lib.lib().lens_seqmatch(_ptr(counts), Q, P, P, L, _ptr(S), P - L + 1, _stream())
    _lib.check(st, "lens_seqmatch")
    return S


def _pair_out(out, shape, device):
    """(int32 idx, float32 val) result buffers of `shape`: the caller's (checked) or fresh ones."""
    if out is None:
        return (torch.empty(shape, dtype=torch.int32, device=device),
                torch.empty(shape, dtype=torch.float32, device=device))
    idx, val = out
    assert idx.dtype == torch.int32 and val.dtype == torch.float32 and idx.is_cuda and val.is_cuda
    assert tuple(idx.shape) == tuple(shape) and tuple(val.shape) == tuple(shape)
    assert idx.is_contiguous() and val.is_contiguous()
    return idx, val


def seqmatch_topk(counts, L, K, n_places=None, place_offset=0, out=None):
    """Sequence matcher + top-K per query (ties -> larger place index). counts [Q, ld] device.
    Returns idx int32 [Q-L+1,K], val float32 [Q-L+1,K] (written into `out` = (idx, val) when given)."""
    _need_cuda(counts, "counts")
    assert counts.dtype == torch.float32 and counts.is_contiguous()
    Q, ld = counts.shape
    if n_places is None:
        n_places = ld - L + 1
    idx, val = _pair_out(out, (Q - L + 1, K), counts.device)
    st = _lib.lib().lens_seqmatch_topk(_ptr(counts), Q, ld, n_places, L, K, place_offset, _ptr(idx), _ptr(val), _stream())
    _lib.check(st, "lens_seqmatch_topk")
    return idx, val


def topk_merge(idx_in, val_in, out=None):
    """Merge per-shard lists [R,Qs,K] -> [Qs,K] with the same ordering rule."""
    _need_cuda(idx_in, "idx_in")
    R, Qs, K = idx_in.shape
    assert idx_in.dtype == torch.int32 and val_in.dtype == torch.float32
    idx_in = idx_in.contiguous(); val_in = val_in.contiguous()
    idx, val = _pair_out(out, (Qs, K), idx_in.device)
    st = _lib.lib().lens_topk_merge(_ptr(idx_in), _ptr(val_in), R, Qs, K, _ptr(idx), _ptr(val), _stream())
    _lib.check(st, "lens_topk_merge")
    return idx, val


# --------------------------------------------------------------------------------------------
# training path (`--train_model`): thin wrappers over lens_layer_forward_clamped / lens_train_layer
# --------------------------------------------------------------------------------------------
def train_in_pad(n_in):
    pad = int(_lib.lib().lens_train_in_pad(int(n_in)))
    if pad == 0:
        raise _lib.LensAmdError(f"training kernel supports a fan-in of at most 2048 inputs, got {n_in}")
    return pad


def pad_table(x, ld):
    """[S, n] float32 (any device) -> zero padded contiguous [S, ld] on the same device."""
    S, n = x.shape
    out = torch.zeros((S, ld), dtype=torch.float32, device=x.device)
    out[:, :n] = x
    return out


def layer_forward_clamped(x_table, n_in, w, thr):
    """clamp(w(x) - thr, 0, 1) for every sample; x_table [S, ld>=n_in] device, w [n_out, n_in], thr [n_out]."""
    _need_cuda(x_table, "x_table")
    S, ldx = x_table.shape
    n_out = w.shape[0]
    y = torch.empty((S, n_out), dtype=torch.float32, device=x_table.device)
    st = _lib.lib().lens_layer_forward_clamped(_ptr(x_table), S, ldx, _ptr(w), w.shape[1], _ptr(thr), n_in, n_out,
                                               _ptr(y), n_out, _stream())
    _lib.check(st, "lens_layer_forward_clamped")
    return y


def train_layer(w, thr, exc, inh, fire_rate, have_rate, x_table, order, eta_ip, eta_stdp, force_idx=None, mpre_table=None):
    """T = len(order) training steps in place on the device tensors w [n_out,n_in] and thr [n_out].
    (The kernel works on zero-padded rows; padding / un-padding happens here, once per call.)"""
    for t, name in ((w, "w"), (thr, "thr"), (x_table, "x_table"), (order, "order")):
        _need_cuda(t, name)
    n_out, n_in = w.shape
    assert thr.is_contiguous() and thr.numel() == n_out
    assert exc.dtype == torch.uint8 and inh.dtype == torch.uint8 and exc.shape == w.shape and inh.shape == w.shape
    assert order.dtype == torch.int32 and eta_ip.dtype == torch.float32 and eta_stdp.dtype == torch.float32
    ld = train_in_pad(n_in)
    assert x_table.shape[1] >= ld
    if force_idx is not None:
        assert force_idx.dtype == torch.int32 and mpre_table is not None and mpre_table.shape == x_table.shape
    wp = pad_table(w, ld)
    ep = torch.zeros((n_out, ld), dtype=torch.uint8, device=w.device); ep[:, :n_in] = exc
    ip = torch.zeros((n_out, ld), dtype=torch.uint8, device=w.device); ip[:, :n_in] = inh
    st = _lib.lib().lens_train_layer(_ptr(wp), ld, _ptr(thr), _ptr(ep), _ptr(ip), ld,
                                     _ptr(fire_rate), int(have_rate), _ptr(x_table), _ptr(mpre_table), x_table.shape[1],
                                     _ptr(order), _ptr(force_idx), _ptr(eta_ip), _ptr(eta_stdp),
                                     n_out, n_in, order.numel(), _stream())
    _lib.check(st, "lens_train_layer")
    w.copy_(wp[:, :n_in])
    return w, thr


# --------------------------------------------------------------------------------------------
# SAD baseline (`--sad`)
# --------------------------------------------------------------------------------------------
def sad_matrix(query, reference):
    """L1 distance of every query frame to every reference frame (lens/src/sad.py:34-38).
    uint8 device tensors [Q,H,W] / [P,H,W] -> float32 [Q,P] (exact integers)."""
    _need_cuda(query, "query"); _need_cuda(reference, "reference")
    assert query.dtype == torch.uint8 and reference.dtype == torch.uint8
    q = query.reshape(query.shape[0], -1).contiguous()
    r = reference.reshape(reference.shape[0], -1).contiguous()
    HW = q.shape[1]
    assert r.shape[1] == HW
    pad = (-HW) % 16                                   # the kernel streams 16-byte words; zero padding adds |0-0|
    if pad:
        q = torch.nn.functional.pad(q, (0, pad)); r = torch.nn.functional.pad(r, (0, pad))
    out = torch.empty((q.shape[0], r.shape[0]), dtype=torch.float32, device=q.device)
    st = _lib.lib().lens_sad_matrix(_ptr(q), q.shape[0], _ptr(r), r.shape[0], q.shape[1], _ptr(out), r.shape[0], _stream())
    _lib.check(st, "lens_sad_matrix")
    return out
