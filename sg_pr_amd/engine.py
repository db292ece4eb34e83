"""ctypes binding of the C-ABI in include/sgpr.h (libsgpr_hip.so) + weight-blob packing.

This is the only place the Python host touches the HIP engine.  PyTorch is used
for device memory and streams only: tensors are handed over as raw pointers.
There is NO CPU fallback: a missing library or a non-GPU tensor raises.
"""
import ctypes
import os

import numpy as np
import torch

from . import _build

NUM_LABELS = 12
F3 = 32
MAX_NODES = 256     # SGPR_MAX_NODES of include/sgpr.h: node_num of the tuned kernels (beyond: the any-shape kernels)

# order of the fp32 tensors in the weights blob (include/sgpr.h, sgpr_weights_count)
_CONV_BLOCKS = ["dgcnn_s_conv1", "dgcnn_f_conv1", "dgcnn_s_conv2", "dgcnn_f_conv2",
                "dgcnn_s_conv3", "dgcnn_f_conv3", "dgcnn_conv_end"]
BLOB_KEYS = []
for _b in _CONV_BLOCKS:
    BLOB_KEYS += [_b + ".0.weight", _b + ".1.weight", _b + ".1.bias", _b + ".1.running_mean", _b + ".1.running_var"]
BLOB_KEYS += ["attention.weight_matrix", "tensor_network.weight_matrix", "tensor_network.weight_matrix_block",
              "tensor_network.bias", "fully_connected_first.weight", "fully_connected_first.bias",
              "scoring_layer.weight", "scoring_layer.bias"]

SGPR_OK = 0
ERROR_NAMES = {-1: "SGPR_E_INVALID", -2: "SGPR_E_DIMS", -3: "SGPR_E_NODES", -4: "SGPR_E_K", -5: "SGPR_E_LABEL",
               -6: "SGPR_E_HIP", -7: "SGPR_E_WORKSPACE", -8: "SGPR_E_BLOB"}

# every symbol include/sgpr.h declares (tests check the library exports all of them)
ABI_SYMBOLS = ["sgpr_weights_count", "sgpr_create", "sgpr_destroy", "sgpr_pooled_width", "sgpr_is_any_shape",
               "sgpr_embed_workspace_bytes", "sgpr_embed",
               "sgpr_embed_capped", "sgpr_embed_ordered", "sgpr_embed_ragged",
               "sgpr_size_order_workspace_bytes", "sgpr_size_order",
               "sgpr_embed_dense", "sgpr_embed_debug", "sgpr_score_pairs", "sgpr_pair_plan_ints", "sgpr_pair_plan",
               "sgpr_score_pair_list_workspace_bytes", "sgpr_score_pair_list", "sgpr_score_all_pairs_workspace_bytes",
               "sgpr_score_all_pairs", "sgpr_score_all_pairs_multi_workspace_bytes", "sgpr_score_all_pairs_multi",
               "sgpr_forward_workspace_bytes", "sgpr_forward_dense", "sgpr_check_status",
               "sgpr_pair_positives", "sgpr_pair_threshold_counts_workspace_bytes", "sgpr_pair_threshold_counts",
               "sgpr_f1_max_workspace_bytes", "sgpr_f1_max", "sgpr_topk_rows",
               "sgpr_embed_lds_bytes", "sgpr_knn", "sgpr_graph_feature", "sgpr_attention_pool", "sgpr_ntn",
               "sgpr_attention_pool_any", "sgpr_ntn_any",
               "sgpr_cluster_workspace_bytes", "sgpr_cluster_scan", "sgpr_graph_edges",
               "sgpr_debug_set_profile_buffer", "sgpr_debug_set_skip_mask", "sgpr_debug_uses_f16_planes", "sgpr_last_error",
               "sgpr_abi_version"]


# struct sgpr_rank_group of include/sgpr.h
RANK_GROUP = np.dtype([("value", "<f4", (8,)), ("pairs", "<u4", (8,))])


class SgprError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("%s (%d): %s" % (ERROR_NAMES.get(code, "SGPR_E_?"), code, message))
        self.code = code


class SgprPairsJob(ctypes.Structure):
    """struct sgpr_pairs_job of include/sgpr.h"""
    _fields_ = [("d_pooled_rows", ctypes.c_void_p), ("R", ctypes.c_int32), ("d_pooled_cols", ctypes.c_void_p),
                ("M", ctypes.c_int32), ("d_score", ctypes.c_void_p), ("ld", ctypes.c_int64)]


class SgprDims(ctypes.Structure):
    _fields_ = [("num_labels", ctypes.c_int32), ("filters_1", ctypes.c_int32), ("filters_2", ctypes.c_int32),
                ("filters_3", ctypes.c_int32), ("tensor_neurons", ctypes.c_int32),
                ("bottle_neck_neurons", ctypes.c_int32)]


_lib = None


def load_library():
    """dlopen libsgpr_hip.so (built in-tree by sg_pr_amd._build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SGPR_HIP_LIB") or _build.LIB_PATH   # override: an alternative build of the same C-ABI
    if not os.path.exists(path):
        raise ImportError("HIP engine %s is not built; run `python -m sg_pr_amd._build` "
                          "(there is no CPU fallback)" % path)
    lib = ctypes.CDLL(path)
    vp, i32, i64, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t
    # first thing: a library built from other sources than the header this binding follows must say so before any
    # missing symbol raises an AttributeError
    want = _build.header_abi_version()
    try:
        lib.sgpr_abi_version.restype = i32
        lib.sgpr_abi_version.argtypes = []
        have = lib.sgpr_abi_version()
    except AttributeError:
        have = None
    if have != want:
        raise ImportError("%s reports C-ABI version %s, include/sgpr.h declares %d: rebuild it "
                          "(`python -m sg_pr_amd._build --force`)" % (path, have, want))
    lib.sgpr_weights_count.restype = sz
    lib.sgpr_weights_count.argtypes = [ctypes.POINTER(SgprDims)]
    lib.sgpr_create.restype = i32
    lib.sgpr_create.argtypes = [vp, sz, ctypes.POINTER(SgprDims), i32, ctypes.POINTER(vp)]
    lib.sgpr_destroy.restype = None
    lib.sgpr_destroy.argtypes = [vp]
    lib.sgpr_pooled_width.restype = i32
    lib.sgpr_pooled_width.argtypes = [vp]
    lib.sgpr_is_any_shape.restype = i32
    lib.sgpr_is_any_shape.argtypes = [vp]
    lib.sgpr_embed_workspace_bytes.restype = sz
    lib.sgpr_embed_workspace_bytes.argtypes = [vp, i32, i32, i32]
    lib.sgpr_embed.restype = i32
    lib.sgpr_embed.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, sz, vp]
    lib.sgpr_embed_capped.restype = i32
    lib.sgpr_embed_capped.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, sz, vp]
    lib.sgpr_embed_ordered.restype = i32
    lib.sgpr_embed_ordered.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, i32, vp, vp, vp, vp, sz, vp]
    lib.sgpr_embed_ragged.restype = i32
    lib.sgpr_embed_ragged.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp, vp, vp, vp, sz, vp]
    lib.sgpr_size_order_workspace_bytes.restype = sz
    lib.sgpr_size_order_workspace_bytes.argtypes = [i32]
    lib.sgpr_size_order.restype = i32
    lib.sgpr_size_order.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, sz, vp]
    lib.sgpr_embed_dense.restype = i32
    lib.sgpr_embed_dense.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, sz, vp]
    lib.sgpr_embed_debug.restype = i32
    lib.sgpr_embed_debug.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.sgpr_score_pairs.restype = i32
    lib.sgpr_score_pairs.argtypes = [vp, vp, vp, vp, vp, i64, vp, vp]
    lib.sgpr_pair_plan_ints.restype = sz
    lib.sgpr_pair_plan_ints.argtypes = [i64, i32]
    lib.sgpr_pair_plan.restype = i32
    lib.sgpr_pair_plan.argtypes = [vp, vp, i64, i32, i32, vp, sz, ctypes.POINTER(sz), ctypes.POINTER(i32),
                                   ctypes.POINTER(i32)]
    lib.sgpr_score_pair_list_workspace_bytes.restype = sz
    lib.sgpr_score_pair_list_workspace_bytes.argtypes = [vp, i32, i32]
    lib.sgpr_score_pair_list.restype = i32
    lib.sgpr_score_pair_list.argtypes = [vp, vp, i32, vp, i32, vp, i32, i32, i64, vp, vp, sz, vp]
    lib.sgpr_score_all_pairs_workspace_bytes.restype = sz
    lib.sgpr_score_all_pairs_workspace_bytes.argtypes = [vp, i32, i32]
    lib.sgpr_score_all_pairs.restype = i32
    lib.sgpr_score_all_pairs.argtypes = [vp, vp, i32, vp, i32, vp, i64, vp, sz, vp]
    lib.sgpr_score_all_pairs_multi_workspace_bytes.restype = sz
    lib.sgpr_score_all_pairs_multi_workspace_bytes.argtypes = [vp, i32, vp]
    lib.sgpr_score_all_pairs_multi.restype = i32
    lib.sgpr_score_all_pairs_multi.argtypes = [vp, i32, vp, vp, sz, vp]
    lib.sgpr_forward_workspace_bytes.restype = sz
    lib.sgpr_forward_workspace_bytes.argtypes = [vp, i32, i32, i32]
    lib.sgpr_forward_dense.restype = i32
    lib.sgpr_forward_dense.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, sz, vp]
    lib.sgpr_check_status.restype = i32
    lib.sgpr_check_status.argtypes = [vp, vp]
    dbl = ctypes.c_double
    lib.sgpr_pair_positives.restype = i32
    lib.sgpr_pair_positives.argtypes = [vp, vp, i32, i32, i64, i32, vp, dbl, dbl, vp, i64, vp, i64, vp, vp]
    lib.sgpr_pair_threshold_counts.restype = i32
    lib.sgpr_pair_threshold_counts_workspace_bytes.restype = sz
    lib.sgpr_pair_threshold_counts_workspace_bytes.argtypes = [vp, i32]
    lib.sgpr_pair_threshold_counts.argtypes = [vp, vp, i32, i32, i64, i32, vp, dbl, dbl, vp, i64, vp, i32, vp, i32, vp, vp, vp, sz,
                                               vp]
    lib.sgpr_f1_max_workspace_bytes.restype = sz
    lib.sgpr_f1_max_workspace_bytes.argtypes = [vp, i32, i32]
    lib.sgpr_f1_max.restype = i32
    lib.sgpr_f1_max.argtypes = [vp, vp, i32, i32, i64, i32, vp, dbl, dbl, vp, i64, vp, vp, sz, vp]
    lib.sgpr_topk_rows.restype = i32
    lib.sgpr_topk_rows.argtypes = [vp, vp, i32, i32, i64, i32, i32, i32, vp, vp, vp]
    lib.sgpr_embed_lds_bytes.restype = sz
    lib.sgpr_embed_lds_bytes.argtypes = [vp, i32, i32]
    lib.sgpr_knn.restype = i32
    lib.sgpr_knn.argtypes = [vp, i32, i32, i32, i32, vp, vp]
    lib.sgpr_graph_feature.restype = i32
    lib.sgpr_graph_feature.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
    lib.sgpr_attention_pool.restype = i32
    lib.sgpr_attention_pool.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    lib.sgpr_ntn.restype = i32
    lib.sgpr_ntn.argtypes = [vp, vp, vp, vp, vp, i64, vp, vp]
    lib.sgpr_attention_pool_any.restype = i32
    lib.sgpr_attention_pool_any.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    lib.sgpr_ntn_any.restype = i32
    lib.sgpr_ntn_any.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, vp, vp]
    lib.sgpr_cluster_workspace_bytes.restype = sz
    lib.sgpr_cluster_workspace_bytes.argtypes = [i32]
    lib.sgpr_cluster_scan.restype = i32
    lib.sgpr_cluster_scan.argtypes = [vp, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.sgpr_graph_edges.restype = i32
    lib.sgpr_graph_edges.argtypes = [vp, i32, vp, i32, i32, vp, vp, vp, sz, vp]
    lib.sgpr_debug_set_skip_mask.restype = None
    lib.sgpr_debug_set_skip_mask.argtypes = [vp, i32]
    lib.sgpr_debug_uses_f16_planes.restype = i32
    lib.sgpr_debug_uses_f16_planes.argtypes = [vp]
    lib.sgpr_debug_set_profile_buffer.restype = None
    lib.sgpr_debug_set_profile_buffer.argtypes = [vp, vp]
    lib.sgpr_last_error.restype = ctypes.c_char_p
    lib.sgpr_last_error.argtypes = []
    _lib = lib
    return lib


def default_dims():
    return SgprDims(NUM_LABELS, 64, 64, 32, 16, 16)


def dims_from_args(args, number_of_labels=NUM_LABELS):
    return SgprDims(int(number_of_labels), int(args.filters_1), int(args.filters_2), int(args.filters_3),
                    int(args.tensor_neurons), int(args.bottle_neck_neurons))


def blob_from_state_dict(sd):
    """Flatten a reference state dict (with or without the `module.` prefix) into the fp32 blob."""
    parts = []
    for key in BLOB_KEYS:
        t = sd[key] if key in sd else sd["module." + key]
        parts.append(t.detach().to(torch.float32).cpu().contiguous().view(-1).numpy())
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.float32)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class Engine:
    """One packed-weights handle on one GPU (immutable after creation)."""

    def __init__(self, state_dict, dims=None, device=0):
        self._h = None
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("sg_pr_amd.Engine needs a ROCm GPU (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.device = torch.device("cuda", int(device) if not isinstance(device, torch.device) else device.index or 0)
        self.dims = dims if dims is not None else default_dims()
        # The tuned kernels run on the built shapes (32 pooled features, 12 label channels ...); a smaller architecture is
        # served by zero-padding its tensors at sgpr_create.  Device buffers keep the built widths (pooled [G, 32]: the
        # channels the checkpoint does not have are exactly 0); node embeddings are cut to filters_3 where they leave the
        # engine.  A larger architecture gets an any-shape handle (plain-fp32 kernels, buffers of the model's own width).
        self.f3 = int(self.dims.filters_3)
        self.pw = F3                                                     # floats per pooled / emb row (set below)
        self.any_shape = False
        if isinstance(state_dict, (np.ndarray, torch.Tensor)):          # already the flat fp32 blob (sg_pr_amd.ops)
            blob = np.ascontiguousarray(torch.as_tensor(state_dict).detach().cpu().numpy(), dtype=np.float32).ravel()
        else:
            blob = blob_from_state_dict(state_dict)
        want = self.lib.sgpr_weights_count(ctypes.byref(self.dims))
        h = ctypes.c_void_p()
        rc = self.lib.sgpr_create(blob.ctypes.data_as(ctypes.c_void_p), blob.size, ctypes.byref(self.dims),
                                  self.device.index, ctypes.byref(h))
        self._check(rc)
        assert want == blob.size
        self._h = h
        self.pw = int(self.lib.sgpr_pooled_width(h))
        self.any_shape = bool(self.lib.sgpr_is_any_shape(h))
        self.num_cus = int(torch.cuda.get_device_properties(self.device).multi_processor_count)
        self._order_cache = []            # (tensor objects, offsets, shape, versions, K) -> device launch order + node_cap, see embed()

    def close(self):
        if self._h is not None:
            self.lib.sgpr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _check(self, rc):
        if rc != SGPR_OK:
            raise SgprError(rc, self.lib.sgpr_last_error().decode())

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, t, dtype, name):
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(t)
        if t.device != self.device or t.dtype != dtype or not t.is_contiguous():
            t = t.to(device=self.device, dtype=dtype).contiguous()
        return t

    def _pooled(self, t, name):
        """a pooled-vector argument on the device, fp32, contiguous - and of THIS handle's row width: the kernels read
        row r at p + r * width, so rows of another engine (or a [:, :filters_3] cut) would be read past their end"""
        t = self._dev(t, torch.float32, name)
        if t.dim() != 2 or t.shape[1] != self.pw:
            raise ValueError("%s must be [rows, %d] (this handle's pooled width), got %s" % (name, self.pw, tuple(t.shape)))
        return t

    def _cut(self, emb):
        """node embeddings [.., pooled width] -> [.., filters_3] (a view; identity for the shipped architecture)"""
        return emb if emb is None or self.f3 == self.pw else emb[..., :self.f3]

    def _ws(self, nbytes):
        return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=self.device)

    def lds_bytes(self, node_num, k):
        return int(self.lib.sgpr_embed_lds_bytes(self._h, node_num, k))

    PHASES = ["stage", "select", "gram", "gemm", "-", "gather", "conv_end", "attention"]

    def phase_profile(self, centers, labels, k, reps=3, node_cap=0, order=None, select_split=False):
        """Debug: workgroup cycles per phase of the embed kernel (thread-0 clocks at the barriers, on the profile
        instance of the kernel).  select_split adds timers inside the selection (they perturb it)."""
        buf = torch.zeros(16, dtype=torch.int64, device=self.device)
        self.lib.sgpr_debug_set_profile_buffer(self._h, _ptr(buf))
        self.lib.sgpr_debug_set_skip_mask(self._h, 128 if select_split else 0)
        try:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                self.embed(centers, labels, k, node_cap=node_cap, order=order)
            e1.record()
            torch.cuda.synchronize(self.device)
            self.last_profile_ms = e0.elapsed_time(e1) / reps     # launch time WITH the timers running
        finally:
            self.lib.sgpr_debug_set_profile_buffer(self._h, None)
            self.lib.sgpr_debug_set_skip_mask(self._h, 0)
        call = buf.cpu().numpy().astype(np.float64)
        self.last_select_split = call[8:14]   # load, sort, merge, tau+masks, prefix, emit (thread-0 cycles)
        c = call[:8]
        return dict(zip(self.PHASES, c / max(c.sum(), 1.0))), c

    def check_status(self):
        """Synchronises the stream and raises SgprError if a launch since the last check saw a label outside
        [-1, num_labels) (the reference raises KeyError, sg_net.py:277) or a graph that broke its node_cap promise."""
        self._check(self.lib.sgpr_check_status(self._h, self._stream()))

    def uses_f16_planes(self):
        """True: the default two-plane f16 datapath serves this checkpoint; False: the wide-range instance throughout."""
        return bool(self.lib.sgpr_debug_uses_f16_planes(self._h))

    def set_skip_mask(self, mask):
        """Debug / ablation only (include/sgpr.h)."""
        self.lib.sgpr_debug_set_skip_mask(self._h, int(mask))

    # ------------------------------------------------------------------ per-graph half
    @staticmethod
    def processed_slots(centers, labels, k):
        """Slots the kernel processes per graph, int64 [G] (see sgpr_embed_capped): trailing slots identical to the
        last one collapse to one representative when there are >= k of them.  numpy arrays or tensors."""
        c = torch.as_tensor(centers)
        l = torch.as_tensor(labels)
        n = l.shape[1]
        same = (l == l[:, -1:]) & (c == c[:, -1:, :]).all(-1)
        idx = torch.arange(n, device=l.device).expand_as(l)
        nd = torch.where(~same, idx + 1, torch.zeros_like(idx)).amax(dim=1)          # slots before the trailing run
        m = n - nd
        return nd + torch.where((m >= k) & (m > 1), torch.ones_like(m), m)

    @staticmethod
    def node_cap_of(centers, labels, k):
        """Largest number of slots the kernel will process for any graph of the batch.
        For device tensors this synchronises (do it once per dataset)."""
        eff = Engine.processed_slots(centers, labels, k)
        return int(eff.max().item()) if eff.numel() else 0

    def size_order(self, centers, labels, k):
        """Launch order for sgpr_embed_ordered: graph indices sorted by processed slots, largest first (stable), plus
        the node_cap they justify -> (order i32 device tensor [G], node_cap).  A property of the packed data - build
        it once per dataset.  Computed on the device (sgpr_size_order: two small launches and a 4-byte read-back, which
        synchronises); host arrays are uploaded first.  Beyond the tuned kernels' node_num / on an any-shape handle
        (where no launch takes a node_cap) the torch form below answers."""
        labels_t = torch.as_tensor(labels)
        g, n = labels_t.shape
        if g == 0:
            return torch.empty(0, dtype=torch.int32, device=self.device), 0
        if n > MAX_NODES or self.any_shape:
            return self.size_order_torch(centers, labels, k)
        order, info = self.size_order_device(self._dev(centers, torch.float32, "centers"),
                                             self._dev(labels_t, torch.int32, "labels"), None, n, k)
        return order, int(info[0].item())

    def size_order_torch(self, centers, labels, k):
        """size_order by torch ops (the checker of the device kernel; any node_num)."""
        eff = self.processed_slots(centers, labels, k)
        if eff.numel() == 0:
            return torch.empty(0, dtype=torch.int32, device=self.device), 0
        order = torch.argsort(eff, descending=True, stable=True).to(torch.int32).to(self.device)
        return order, int(eff.max().item())

    def size_order_device(self, centers, labels, offsets, node_num, k):
        """sgpr_size_order without the read-back: (order i32 [G], info i32 [2] = node_cap, graphs beyond 64 slots), both
        on the device, asynchronous.  Padded arrays (offsets None) or a ragged store's offsets (centers / labels None)."""
        g = (offsets.numel() - 1) if offsets is not None else labels.shape[0]
        order = torch.empty(g, dtype=torch.int32, device=self.device)
        info = torch.empty(2, dtype=torch.int32, device=self.device)      # (both entries are written by the kernel)
        ws_bytes = self.lib.sgpr_size_order_workspace_bytes(g)
        ws = self._ws(ws_bytes)
        rc = self.lib.sgpr_size_order(self._h, _ptr(centers if offsets is None else None),
                                      _ptr(labels if offsets is None else None), _ptr(offsets), g, int(node_num), int(k),
                                      _ptr(order), _ptr(info), _ptr(ws), ws_bytes, self._stream())
        self._check(rc)
        return order, info

    def _cached_order(self, centers, labels, k):
        """The largest-first launch order and the node_cap of a RESIDENT batch (sgpr_size_order, asynchronous), remembered per
        tensor pair, so that evaluating the same packed store again costs nothing -> (order, node_cap or 0).  An entry
        belongs to the tensors' base OBJECTS (weak references: a new tensor that happens to reuse the address of a freed
        one is a different object), their storage offsets / shapes, torch's in-place version counters and K.  The order is
        used at once; the node_cap travels to pinned host memory behind an event and is used from the first later call
        that finds the copy complete (no synchronisation, ever).  Whatever the cache says, the kernels check it: an order
        is a permutation of the batch's graphs (it decides when a graph runs, never what it yields), a node_cap a
        promise (a graph beyond it: NaN + SGPR_E_NODES, never a wrong result).  A data-set property kept by the binding;
        the C-ABI stays stateless."""
        import weakref
        bc = centers._base if centers._base is not None else centers
        bl = labels._base if labels._base is not None else labels
        key = (centers.storage_offset(), labels.storage_offset(), tuple(labels.shape), centers._version, labels._version, int(k))
        live = []
        hit = None
        for rc, rl, kk, order, info_h, ev in self._order_cache:
            if rc() is None or rl() is None:
                continue                                   # a tensor of the entry is gone
            live.append((rc, rl, kk, order, info_h, ev))
            if kk == key and rc() is bc and rl() is bl:
                hit = (order, int(info_h[0]) if ev.query() else 0)
        self._order_cache = live
        if hit is not None:
            return hit
        order, info = self.size_order_device(centers, labels, None, labels.shape[1], k)
        info_h = torch.empty(2, dtype=torch.int32).pin_memory()
        info_h.copy_(info, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._order_cache = self._order_cache[-7:] + [(weakref.ref(bc), weakref.ref(bl), key, order, info_h, ev)]
        return order, 0

    def embed(self, centers, labels, k, want_att=False, want_emb=False, debug=False, node_cap=0, order=None, auto_order=True):
        """centers [G,N,3] f32, labels [G,N] i32 (-1 = pad) -> pooled [G,32] (+ att [G,N], emb [G,N,32]).
        node_cap: optional promise on the processed slots per graph (node_cap_of); 0 = none.
        order: optional i32 launch order (size_order) - graphs not listed keep uninitialised output rows.
        auto_order: no order given and the arrays already resident on this device (more graphs than CUs, the tuned kernels'
        node_num): launch in the largest-first order sgpr_size_order makes on the device, computed once per tensor pair
        (_cached_order) - same bits, ~5 % faster than storage order on KITTI-like data; from the second call on the batch's
        node_cap (made by the same kernels) is promised as well; False = the plain C-ABI call."""
        resident = (isinstance(centers, torch.Tensor) and isinstance(labels, torch.Tensor) and centers.device == self.device
                    and labels.device == self.device and centers.dtype == torch.float32 and labels.dtype == torch.int32
                    and centers.is_contiguous() and labels.is_contiguous())
        centers = self._dev(centers, torch.float32, "centers")
        labels = self._dev(labels, torch.int32, "labels")
        g, n = labels.shape
        assert centers.shape == (g, n, 3), "centers must be [G, N, 3]"
        if (order is None and auto_order and resident and not debug and g > self.num_cus and n <= MAX_NODES and k <= n
                and not self.any_shape):
            order, cap = self._cached_order(centers, labels, k)
            if node_cap == 0 and 0 < cap < n:
                node_cap = cap
        pooled = torch.empty(g, self.pw, dtype=torch.float32, device=self.device)
        att = torch.empty(g, n, dtype=torch.float32, device=self.device) if (want_att or debug) else None
        emb = torch.empty(g, n, self.pw, dtype=torch.float32, device=self.device) if (want_emb or debug) else None
        ws_bytes = self.lib.sgpr_embed_workspace_bytes(self._h, g, n, k)
        ws = self._ws(ws_bytes)
        if debug:
            layers = torch.zeros(g, 6, n, 64, dtype=torch.float32, device=self.device)
            knn = torch.full((g, 6, n, k), -1, dtype=torch.int32, device=self.device)
            rc = self.lib.sgpr_embed_debug(self._h, _ptr(centers), _ptr(labels), g, n, k, _ptr(pooled), _ptr(att),
                                           _ptr(emb), _ptr(layers), _ptr(knn), _ptr(ws), ws_bytes, self._stream())
            self._check(rc)
            return pooled, att, self._cut(emb), layers, knn
        if g == 0:
            return pooled, att, self._cut(emb)
        if order is not None:
            order = self._dev(order, torch.int32, "order")
            rc = self.lib.sgpr_embed_ordered(self._h, _ptr(centers), _ptr(labels), g, n, int(node_cap), k, _ptr(order),
                                             order.numel(), _ptr(pooled), _ptr(att), _ptr(emb), _ptr(ws), ws_bytes,
                                             self._stream())
            self._check(rc)
            return pooled, att, self._cut(emb)
        rc = self.lib.sgpr_embed_capped(self._h, _ptr(centers), _ptr(labels), g, n, int(node_cap), k, _ptr(pooled),
                                        _ptr(att), _ptr(emb), _ptr(ws), ws_bytes, self._stream())
        self._check(rc)
        return pooled, att, self._cut(emb)

    @staticmethod
    def to_ragged(centers, labels, num_labels=NUM_LABELS):
        """Padded arrays (centers [G,N,3], labels [G,N], -1 = pad, padding trailing) -> the ragged store of
        sgpr_embed_ragged: (centers f32 [S,3], labels i8 [S], offsets i64 [G+1]) as numpy arrays."""
        import numpy as np
        c = np.asarray(centers, dtype=np.float32)
        l = np.asarray(labels)
        real = l >= 0
        if real.any() and int(l[real].max()) >= num_labels:
            # an int8 cast would wrap 256 + c into the valid class c: refuse like the padded path and the reference do
            # (KeyError, sg_net.py:277)
            raise ValueError("to_ragged: label %d outside [0, %d)" % (int(l[real].max()), num_labels))
        counts = real.sum(1)
        if not (real == (np.arange(l.shape[1])[None, :] < counts[:, None])).all():
            raise ValueError("to_ragged: padding slots (label -1) must trail the real nodes of every graph")
        offsets = np.zeros(l.shape[0] + 1, dtype=np.int64)
        np.cumsum(counts, out=offsets[1:])
        return np.ascontiguousarray(c[real]), np.ascontiguousarray(l[real].astype(np.int8)), offsets

    @staticmethod
    def ragged_blob(centers, labels, offsets, pin=True):
        """The ragged store as ONE host buffer (offsets | centers | labels, each part 16-byte aligned): one H2D copy
        instead of three (each copy has its own ~10 us of submission).  -> (uint8 tensor, pinned when a GPU is there, layout)."""
        import numpy as np
        parts = (np.ascontiguousarray(offsets, dtype=np.int64), np.ascontiguousarray(centers, dtype=np.float32),
                 np.ascontiguousarray(labels, dtype=np.int8))
        starts, at = [], 0
        for x in parts:
            starts.append(at)
            at = (at + x.nbytes + 15) & ~15
        blob = torch.empty(max(at, 16), dtype=torch.uint8)
        if pin and torch.cuda.is_available():
            blob = blob.pin_memory()
        for x, st in zip(parts, starts):
            blob[st:st + x.nbytes] = torch.from_numpy(x.reshape(-1).view(np.uint8))
        layout = {"offsets": (starts[0], parts[0].shape[0]), "centers": (starts[1], parts[1].shape[0]),
                  "labels": (starts[2], parts[2].shape[0])}
        return blob, layout

    @staticmethod
    def ragged_views(blob, layout):
        """(centers f32 [S,3], labels i8 [S], offsets i64 [G+1]) as views of a blob of ragged_blob - on whatever device
        the blob lives (the arguments of embed_ragged)."""
        o0, on = layout["offsets"]
        c0, cn = layout["centers"]
        l0, ln = layout["labels"]
        return (blob[c0:c0 + cn * 12].view(torch.float32).view(cn, 3), blob[l0:l0 + ln].view(torch.int8),
                blob[o0:o0 + on * 8].view(torch.int64))

    def ragged_order(self, offsets, node_num, k):
        """size_order for a ragged store: (largest-first launch order i32 device tensor, node_cap).  A graph of c nodes
        in node_num slots has m = node_num - c padding slots, of which one is processed when m >= k (else all m)."""
        if node_num <= MAX_NODES and not self.any_shape and len(offsets) > 1:
            order, info = self.size_order_device(None, None, self._dev(offsets, torch.int64, "offsets"), node_num, k)
            return order, int(info[0].item())
        off = torch.as_tensor(offsets).to(torch.int64).cpu()
        cnt = off[1:] - off[:-1]
        m = node_num - cnt
        eff = cnt + torch.where((m >= k) & (m > 1), torch.ones_like(m), m)
        if eff.numel() == 0:
            return torch.empty(0, dtype=torch.int32, device=self.device), 0
        order = torch.argsort(eff, descending=True, stable=True).to(torch.int32).to(self.device)
        return order, int(eff.max().item())

    def embed_ragged(self, centers, labels, offsets, node_num, k, want_att=False, want_emb=False, node_cap=0, order=None):
        """Ragged store (to_ragged) -> pooled [G,32] (+ att [G,node_num], emb [G,node_num,32]); bit-identical to `embed`
        on the padded arrays.  node_cap / order: see ragged_order."""
        centers = self._dev(centers, torch.float32, "centers")
        labels = self._dev(labels, torch.int8, "labels")
        offsets = self._dev(offsets, torch.int64, "offsets")
        g, n = offsets.numel() - 1, int(node_num)
        assert centers.dim() == 2 and centers.shape[1] == 3 and labels.shape[0] == centers.shape[0], "centers [S,3], labels [S]"
        pooled = torch.empty(g, self.pw, dtype=torch.float32, device=self.device)
        att = torch.empty(g, n, dtype=torch.float32, device=self.device) if want_att else None
        emb = torch.empty(g, n, self.pw, dtype=torch.float32, device=self.device) if want_emb else None
        if g == 0:
            return pooled, att, emb
        ws_bytes = self.lib.sgpr_embed_workspace_bytes(self._h, g, n, k)
        ws = self._ws(ws_bytes)
        if order is not None:
            order = self._dev(order, torch.int32, "order")
        rc = self.lib.sgpr_embed_ragged(self._h, _ptr(centers), _ptr(labels), _ptr(offsets), g, n, int(node_cap), k,
                                        _ptr(order), order.numel() if order is not None else 0, _ptr(pooled), _ptr(att),
                                        _ptr(emb), _ptr(ws), ws_bytes, self._stream())
        self._check(rc)
        return pooled, att, self._cut(emb)

    def embed_dense(self, features, k, want_att=False, want_emb=False):
        """features [G, 3+L, N] f32 (the reference's dense layout) -> pooled (+ att, emb)."""
        features = self._dev(features, torch.float32, "features")
        g, ch, n = features.shape
        if ch != 3 + self.dims.num_labels:
            raise ValueError("features must be [G, %d, N], got %s" % (3 + self.dims.num_labels, tuple(features.shape)))
        pooled = torch.empty(g, self.pw, dtype=torch.float32, device=self.device)
        att = torch.empty(g, n, dtype=torch.float32, device=self.device) if want_att else None
        emb = torch.empty(g, n, self.pw, dtype=torch.float32, device=self.device) if want_emb else None
        ws_bytes = self.lib.sgpr_embed_workspace_bytes(self._h, g, n, k)
        ws = self._ws(ws_bytes)
        rc = self.lib.sgpr_embed_dense(self._h, _ptr(features), g, n, k, _ptr(pooled), _ptr(att), _ptr(emb), _ptr(ws),
                                       ws_bytes, self._stream())
        self._check(rc)
        return pooled, att, self._cut(emb)

    # ------------------------------------------------------------------ pair-coupled half
    def score_pairs(self, pooled1, pooled2, idx1=None, idx2=None, out=None):
        pooled1 = self._pooled(pooled1, "pooled1")
        pooled2 = self._pooled(pooled2, "pooled2")
        if idx1 is not None:
            idx1 = self._dev(idx1, torch.int32, "idx1")
        if idx2 is not None:
            idx2 = self._dev(idx2, torch.int32, "idx2")
        n = idx1.numel() if idx1 is not None else pooled1.shape[0]
        n2 = idx2.numel() if idx2 is not None else pooled2.shape[0]
        if n != n2:
            raise ValueError("pair sides differ in length: %d vs %d" % (n, n2))
        score = out if out is not None else torch.empty(n, dtype=torch.float32, device=self.device)
        rc = self.lib.sgpr_score_pairs(self._h, _ptr(pooled1), _ptr(idx1), _ptr(pooled2), _ptr(idx2), n, _ptr(score),
                                       self._stream())
        self._check(rc)
        return score

    def pair_plan(self, idx1, idx2, num_rows, num_cols):
        """Group a pair list by row graph for score_pair_list (sgpr_pair_plan; host work, once per list - like a launch
        order).  idx1 / idx2: integer arrays on the host (pair p = (idx1[p], idx2[p])).  Returns a PairPlan."""
        return PairPlan(self, idx1, idx2, num_rows, num_cols)

    def score_pair_list(self, pooled_rows, pooled_cols, plan, out=None):
        """score[p] = SG-tail(pooled_rows[idx1[p]], pooled_cols[idx2[p]]) for the pairs of `plan` (sgpr_score_pair_list):
        the bilinear form hoisted per distinct row graph, a row's listed columns through the matrix cores 16 at a time;
        bit-identical to score_all_pairs' entries at the listed indices."""
        rows = self._pooled(pooled_rows, "pooled_rows")
        cols = self._pooled(pooled_cols, "pooled_cols")
        if rows.shape[0] != plan.num_rows or cols.shape[0] != plan.num_cols:
            raise ValueError("plan was built for %d x %d graphs, got %d x %d"
                             % (plan.num_rows, plan.num_cols, rows.shape[0], cols.shape[0]))
        score = out if out is not None else torch.empty(plan.P, dtype=torch.float32, device=self.device)
        assert score.numel() == plan.P and score.is_contiguous()
        if plan.P == 0:
            return score
        ws_bytes = self.lib.sgpr_score_pair_list_workspace_bytes(self._h, plan.n_rows, plan.num_cols)
        ws = self._ws(ws_bytes)
        rc = self.lib.sgpr_score_pair_list(self._h, _ptr(rows), plan.num_rows, _ptr(cols), plan.num_cols, _ptr(plan.words),
                                           plan.n_rows, plan.n_items, plan.P, _ptr(score), _ptr(ws), ws_bytes,
                                           self._stream())
        self._check(rc)
        return score

    def score_all_pairs(self, pooled_rows, pooled_cols, out=None):
        rows = self._pooled(pooled_rows, "pooled_rows")
        cols = self._pooled(pooled_cols, "pooled_cols")
        r, m = rows.shape[0], cols.shape[0]
        score = out if out is not None else torch.empty(r, m, dtype=torch.float32, device=self.device)
        assert score.shape == (r, m) and score.stride(1) == 1
        ws_bytes = self.lib.sgpr_score_all_pairs_workspace_bytes(self._h, r, m)
        ws = self._ws(ws_bytes)
        rc = self.lib.sgpr_score_all_pairs(self._h, _ptr(rows), r, _ptr(cols), m, _ptr(score), score.stride(0),
                                           _ptr(ws), ws_bytes, self._stream())
        self._check(rc)
        return score

    MAX_PAIR_JOBS = 8

    def score_all_pairs_multi(self, jobs):
        """Several independent rectangles with one pair of launches (sgpr_score_all_pairs_multi).  jobs: list of
        (pooled_rows, pooled_cols) or (pooled_rows, pooled_cols, out) -> list of [R, M] score tensors."""
        outs, keep, descr = [], [], []
        for job in jobs:
            rows = self._pooled(job[0], "pooled_rows")
            cols = self._pooled(job[1], "pooled_cols")
            out = job[2] if len(job) > 2 and job[2] is not None else torch.empty(rows.shape[0], cols.shape[0],
                                                                                   dtype=torch.float32, device=self.device)
            assert out.shape == (rows.shape[0], cols.shape[0]) and out.stride(1) == 1
            keep.append((rows, cols))
            outs.append(out)
            descr.append(SgprPairsJob(rows.data_ptr(), rows.shape[0], cols.data_ptr(), cols.shape[0], out.data_ptr(),
                                      out.stride(0) if out.shape[0] > 1 else max(out.shape[1], 1)))
        for i in range(0, len(descr), self.MAX_PAIR_JOBS):
            part = descr[i:i + self.MAX_PAIR_JOBS]
            arr = (SgprPairsJob * len(part))(*part)
            ws_bytes = self.lib.sgpr_score_all_pairs_multi_workspace_bytes(self._h, len(part), arr)
            ws = self._ws(ws_bytes)
            rc = self.lib.sgpr_score_all_pairs_multi(self._h, len(part), arr, _ptr(ws), ws_bytes, self._stream())
            self._check(rc)
        return outs

    # ------------------------------------------------------------------ consumers of the score matrix
    def _truth(self, score, row0, pose_xz, gt):
        if not isinstance(score, torch.Tensor):
            score = torch.as_tensor(score)
        if score.device != self.device or score.dtype != torch.float32 or score.dim() != 2 or score.stride(1) != 1:
            score = score.to(device=self.device, dtype=torch.float32).contiguous()      # (row-strided views pass as they are)
        r, m = score.shape
        if pose_xz is not None:
            pose_xz = self._dev(pose_xz, torch.float64, "pose_xz")
            assert pose_xz.shape[1] == 2 and pose_xz.shape[0] >= max(m, row0 + r)
            gt = None
        elif gt is not None:
            gt = self._dev(gt, torch.int8, "gt")
            assert gt.shape == (r, m)
        else:
            raise ValueError("the pair consumers need poses or explicit labels")
        return score, r, m, pose_xz, gt

    def pair_positives(self, score, row0=0, pose_xz=None, d_pos=3.0, d_neg=20.0, gt=None):
        """Scores of the positive pairs of a rectangle that stays on the device (sgpr_pair_positives): float32 device
        tensor (unordered) and the number of positives skipped for a negative / NaN score."""
        score, r, m, pose_xz, gt = self._truth(score, row0, pose_xz, gt)
        count = torch.empty(2, dtype=torch.int64, device=self.device)
        # one launch when the list fits the first guess (positives are rare: loop closures), a second one sized exactly
        # otherwise
        cap = min(r * m, 1 << 20)
        while True:
            out = torch.empty(cap, dtype=torch.float32, device=self.device)
            rc = self.lib.sgpr_pair_positives(self._h, _ptr(score), r, m, score.stride(0), int(row0), _ptr(pose_xz),
                                              float(d_pos), float(d_neg), _ptr(gt), m, _ptr(out) if cap else None, cap,
                                              _ptr(count), self._stream())
            self._check(rc)
            n, bad = (int(v) for v in count.tolist())
            if n <= cap:
                return out[:n], bad
            cap = n

    def pair_threshold_counts(self, score, thresholds, row0=0, pose_xz=None, d_pos=3.0, d_neg=20.0, gt=None, rank=None):
        """One streaming pass over a score rectangle (sgpr_pair_threshold_counts): negatives by threshold bucket.
        thresholds: ascending float32 (<= 8191).  rank = (values, step, above) additionally ranks every negative among
        all distinct positive values (see include/sgpr.h).  Returns (counts int64 [T+1], skipped, rank_sum or None)."""
        score, r, m, pose_xz, gt = self._truth(score, row0, pose_xz, gt)
        thr = self._dev(torch.as_tensor(np.ascontiguousarray(thresholds, dtype=np.float32)), torch.float32, "thresholds")
        t = int(thr.numel())
        out = torch.empty(t + 3, dtype=torch.int64, device=self.device)
        ws_bytes = self.lib.sgpr_pair_threshold_counts_workspace_bytes(self._h, t)
        ws = self._ws(ws_bytes)
        table, at_least, gpt = None, None, 0
        if rank is not None:
            vals, step, above = rank
            vals = np.asarray(vals, dtype=np.float32)
            above = np.asarray(above, dtype=np.int64)
            u, step = int(vals.size), int(step)
            assert above.size == u + 1 and t == -(-u // step) and int(above[0]) < 2 ** 32
            gpt = -(-step // 8)
            # values / pair counts of threshold q's bucket = entries q * step .. (q + 1) * step, padded to gpt * 8
            pad = t * step - u
            v2 = np.concatenate((vals, np.full(pad, np.inf, dtype=np.float32))).reshape(t, step)
            m2 = np.concatenate((above[:-1] - above[1:], np.zeros(pad, dtype=np.int64))).reshape(t, step)
            v3 = np.full((t, gpt * 8), np.inf, dtype=np.float32)
            m3 = np.zeros((t, gpt * 8), dtype=np.uint32)
            v3[:, :step] = v2
            m3[:, :step] = m2
            ent = np.zeros((t, gpt), dtype=RANK_GROUP)
            ent["value"] = v3.reshape(t, gpt, 8)
            ent["pairs"] = m3.reshape(t, gpt, 8)
            table = torch.from_numpy(ent.view(np.uint8).reshape(-1)).to(self.device)
            at_least = torch.from_numpy(np.ascontiguousarray(above[:-1][::step])).to(self.device)
        rc = self.lib.sgpr_pair_threshold_counts(self._h, _ptr(score), r, m, score.stride(0), int(row0), _ptr(pose_xz),
                                                 float(d_pos), float(d_neg), _ptr(gt), m, _ptr(thr), t, _ptr(table), gpt,
                                                 _ptr(at_least), _ptr(out), _ptr(ws), ws_bytes, self._stream())
        self._check(rc)
        h = out.cpu().numpy()
        rank_sum = int(h[t + 2].astype(np.uint64)) if rank is not None else None
        return h[:t + 1].copy(), int(h[t + 1]), rank_sum

    def f1_max(self, score, row0=0, pose_xz=None, d_pos=3.0, d_neg=20.0, gt=None):
        """F1-max of a score rectangle in ONE engine call (sgpr_f1_max): every step on the device, one 64-byte copy
        at the end.  Returns the raw result vector (numpy float64 [8], see include/sgpr.h): [0] F1-max, [1] status (0 ok,
        1 = use the multi-call path, 2 = negative / NaN scores), [2] positives, [3] negatives, [4] passes."""
        score, r, m, pose_xz, gt = self._truth(score, row0, pose_xz, gt)
        res = torch.empty(8, dtype=torch.float64, device=self.device)
        ws_bytes = self.lib.sgpr_f1_max_workspace_bytes(self._h, r, m)
        ws = self._ws(ws_bytes)
        rc = self.lib.sgpr_f1_max(self._h, _ptr(score), r, m, score.stride(0), int(row0), _ptr(pose_xz), float(d_pos),
                                  float(d_neg), _ptr(gt), m, _ptr(res), _ptr(ws), ws_bytes, self._stream())
        self._check(rc)
        # (the 64 bytes land in a pinned buffer of this engine: a pageable copy is staged by the runtime)
        host = getattr(self, "_f1_host", None)
        if host is None:
            host = self._f1_host = torch.empty(8, dtype=torch.float64).pin_memory()
        host.copy_(res, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return host.numpy().copy()

    def topk_rows(self, score, k=1, row0=0, window=-1):
        """Best k columns per row outside |col - (row0 + row)| <= window -> (values f32 [R,k], indices i32 [R,k])."""
        score = self._dev(score, torch.float32, "score")
        r, m = score.shape
        assert score.stride(1) == 1
        vals = torch.empty(r, k, dtype=torch.float32, device=self.device)
        idx = torch.empty(r, k, dtype=torch.int32, device=self.device)
        rc = self.lib.sgpr_topk_rows(self._h, _ptr(score), r, m, score.stride(0), int(row0), int(window), int(k),
                                     _ptr(vals), _ptr(idx), self._stream())
        self._check(rc)
        return vals, idx

    def forward_dense(self, features_1, features_2, k, want_att=True):
        """Drop-in SG.forward on dense [B,3+L,N] inputs -> (score [B], att1 [B,N], att2 [B,N])."""
        f1 = self._dev(features_1, torch.float32, "features_1")
        f2 = self._dev(features_2, torch.float32, "features_2")
        if f1.shape != f2.shape:
            raise ValueError("features_1 / features_2 shapes differ")
        b, ch, n = f1.shape
        if ch != 3 + self.dims.num_labels:
            raise ValueError("features must be [B, %d, N]" % (3 + self.dims.num_labels))
        score = torch.empty(b, dtype=torch.float32, device=self.device)
        att = torch.empty(2, b, n, dtype=torch.float32, device=self.device) if want_att else None
        ws_bytes = self.lib.sgpr_forward_workspace_bytes(self._h, b, n, k)
        ws = self._ws(ws_bytes)
        rc = self.lib.sgpr_forward_dense(self._h, _ptr(f1), _ptr(f2), b, n, k, _ptr(score),
                                         _ptr(att[0]) if want_att else None, _ptr(att[1]) if want_att else None,
                                         _ptr(ws), ws_bytes, self._stream())
        self._check(rc)
        if want_att:
            return score, att[0], att[1]
        return score, None, None


class PairPlan:
    """A pair list grouped by row graph (include/sgpr.h, sgpr_pair_plan): built on the host once, kept on the device."""

    def __init__(self, engine, idx1, idx2, num_rows, num_cols):
        i1 = np.ascontiguousarray(np.asarray(idx1).reshape(-1), dtype=np.int32)
        i2 = np.ascontiguousarray(np.asarray(idx2).reshape(-1), dtype=np.int32)
        if i1.size != i2.size:
            raise ValueError("pair sides differ in length: %d vs %d" % (i1.size, i2.size))
        lib = engine.lib
        self.P, self.num_rows, self.num_cols = int(i1.size), int(num_rows), int(num_cols)
        cap = int(lib.sgpr_pair_plan_ints(self.P, self.num_rows))
        words = np.empty(max(cap, 1), dtype=np.int32)
        used, nr, ni = ctypes.c_size_t(), ctypes.c_int32(), ctypes.c_int32()
        rc = lib.sgpr_pair_plan(i1.ctypes.data_as(ctypes.c_void_p), i2.ctypes.data_as(ctypes.c_void_p), self.P,
                                self.num_rows, self.num_cols, words.ctypes.data_as(ctypes.c_void_p), words.size,
                                ctypes.byref(used), ctypes.byref(nr), ctypes.byref(ni))
        if rc != SGPR_OK:
            raise SgprError(rc, lib.sgpr_last_error().decode())
        self.n_rows, self.n_items = int(nr.value), int(ni.value)
        self.host_words = words[:int(used.value)]
        self.words = torch.from_numpy(self.host_words.copy()).to(engine.device)


# ---------------------------------------------------------------------- handle-free entry points (stand-alone modules)
def _gpu_f32(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a tensor on the MI355X (there is no CPU fallback)" % name)
    return t.detach().to(torch.float32).contiguous()


def _raise_if(lib, rc):
    if rc != SGPR_OK:
        raise SgprError(rc, lib.sgpr_last_error().decode())


def _stream_of(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def knn(x, k):
    """dgcnn.knn (dgcnn.py:14-20): x [B,C,N] on the GPU -> idx [B,N,k] int64."""
    lib = load_library()
    x = _gpu_f32(x, "x")
    b, c, n = x.shape
    idx = torch.empty(b, n, int(k), dtype=torch.int64, device=x.device)
    with torch.cuda.device(x.device):
        _raise_if(lib, lib.sgpr_knn(_ptr(x), b, c, n, int(k), _ptr(idx), _stream_of(x)))
    return idx


def graph_feature(x, idx):
    """dgcnn.get_graph_feature's gather (dgcnn.py:30-49): x [B,C,N], idx [B,N,k] int64 -> [B,2C,N,k]."""
    lib = load_library()
    x = _gpu_f32(x, "x")
    b, c, n = x.shape
    idx = idx.to(device=x.device, dtype=torch.int64).contiguous()
    if idx.shape[:2] != (b, n):
        raise ValueError("idx must be [B, N, k], got %s" % (tuple(idx.shape),))
    k = idx.shape[2]
    out = torch.empty(b, 2 * c, n, k, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _raise_if(lib, lib.sgpr_graph_feature(_ptr(x), _ptr(idx), b, c, n, k, _ptr(out), _stream_of(x)))
    return out


def attention_pool(weight, emb):
    """AttentionModule.forward (layers_batch.py:28-39): weight [32,32], emb [B,N,32] -> (rep [B,32], att [B,N])."""
    lib = load_library()
    emb = _gpu_f32(emb, "embedding")
    weight = _gpu_f32(weight, "weight_matrix").to(emb.device)
    b, n, f = emb.shape
    if tuple(weight.shape) != (f, f):
        raise ValueError("attention_pool: weight_matrix must be [%d, %d]" % (f, f))
    if f > F3:          # wider than the built module: the any-width kernel (plain fp32)
        rep = torch.empty(b, f, dtype=torch.float32, device=emb.device)
        att = torch.empty(b, n, dtype=torch.float32, device=emb.device)
        with torch.cuda.device(emb.device):
            _raise_if(lib, lib.sgpr_attention_pool_any(_ptr(weight), _ptr(emb), b, n, f, _ptr(rep), _ptr(att), _stream_of(emb)))
        return rep, att
    if f < F3:          # a smaller width is the built one with zero channels (exactly: zeros add nothing to any sum)
        emb = torch.nn.functional.pad(emb, (0, F3 - f))
        weight = torch.nn.functional.pad(weight, (0, F3 - f, 0, F3 - f))
    rep = torch.empty(b, F3, dtype=torch.float32, device=emb.device)
    att = torch.empty(b, n, dtype=torch.float32, device=emb.device)
    with torch.cuda.device(emb.device):
        _raise_if(lib, lib.sgpr_attention_pool(_ptr(weight), _ptr(emb), b, n, _ptr(rep), _ptr(att), _stream_of(emb)))
    return rep[:, :f], att


def ntn(weight, weight_block, bias, e1, e2):
    """TenorNetworkModule.forward (layers_batch.py:70-83): e1, e2 [B,32] -> [B,16]."""
    lib = load_library()
    e1 = _gpu_f32(e1, "embedding_1")
    e2 = _gpu_f32(e2, "embedding_2").to(e1.device)
    w = _gpu_f32(weight, "weight_matrix").to(e1.device)
    wb = _gpu_f32(weight_block, "weight_matrix_block").to(e1.device)
    bs = _gpu_f32(bias, "bias").to(e1.device).view(-1)
    b = e1.shape[0]
    f, t = (int(w.shape[0]), int(w.shape[2])) if w.dim() == 3 else (-1, -1)
    if (f < 1 or t < 1 or e1.shape != (b, f) or e2.shape != (b, f) or tuple(w.shape) != (f, f, t) or
            tuple(wb.shape) != (t, 2 * f) or bs.numel() != t):
        raise ValueError("ntn: weight_matrix [F,F,T], weight_matrix_block [T,2F], bias [T], embeddings [B,F]")
    if f > F3 or t > 16:   # wider than the built module: the any-width kernel (plain fp32)
        out = torch.empty(b, t, dtype=torch.float32, device=e1.device)
        with torch.cuda.device(e1.device):
            _raise_if(lib, lib.sgpr_ntn_any(_ptr(w.contiguous()), _ptr(wb.contiguous()), _ptr(bs.contiguous()), _ptr(e1),
                                            _ptr(e2), b, f, t, _ptr(out), _stream_of(e1)))
        return out
    if f < F3 or t < 16:   # a smaller module is the built one with zero weights for what it does not have
        pad = torch.nn.functional.pad
        e1, e2 = pad(e1, (0, F3 - f)).contiguous(), pad(e2, (0, F3 - f)).contiguous()
        w = pad(w, (0, 16 - t, 0, F3 - f, 0, F3 - f)).contiguous()
        wb = pad(torch.cat((pad(wb[:, :f], (0, F3 - f)), pad(wb[:, f:], (0, F3 - f))), dim=1), (0, 0, 0, 16 - t)).contiguous()
        bs = pad(bs, (0, 16 - t)).contiguous()
    out = torch.empty(b, 16, dtype=torch.float32, device=e1.device)
    with torch.cuda.device(e1.device):
        _raise_if(lib, lib.sgpr_ntn(_ptr(w), _ptr(wb), _ptr(bs), _ptr(e1), _ptr(e2), b, _ptr(out), _stream_of(e1)))
    return out[:, :t]


def cluster_scan(points, labels, max_nodes=1024, want_point_node=False):
    """sgpr_cluster_scan: points [P, >=3] f32 and raw labels [P] (u32 bit pattern) on the GPU ->
    (centers float64 [n,3], node labels int32 [n], cluster sizes int32 [n], point -> node int32 [P] or None)."""
    lib = load_library()
    if not isinstance(points, torch.Tensor) or not points.is_cuda:
        raise RuntimeError("points must be a tensor on the MI355X (there is no CPU fallback)")
    points = points.detach().to(torch.float32).contiguous()
    p, stride = points.shape
    labels = labels.to(points.device).contiguous()
    if labels.dtype not in (torch.int32, torch.uint32) or labels.numel() != p:
        raise ValueError("labels must be [P] int32 / uint32 (the raw .label words)")
    dev = points.device
    centers = torch.zeros(max_nodes, 3, dtype=torch.float64, device=dev)
    nlab = torch.zeros(max_nodes, dtype=torch.int32, device=dev)
    nsize = torch.zeros(max_nodes, dtype=torch.int32, device=dev)
    pnode = torch.empty(p, dtype=torch.int32, device=dev) if want_point_node else None
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    ws_bytes = lib.sgpr_cluster_workspace_bytes(p)
    ws = torch.empty(max(int(ws_bytes), 16), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _raise_if(lib, lib.sgpr_cluster_scan(_ptr(points), stride, _ptr(labels), p, max_nodes, _ptr(centers), _ptr(nlab),
                                             _ptr(nsize), _ptr(pnode), _ptr(count), _ptr(ws), ws_bytes, _stream_of(points)))
    n = int(count.item())
    if n < 0 or n > max_nodes:
        raise SgprError(-3, "scan produced %s nodes, more than max_nodes=%d" % ("> 8192" if n < 0 else n, max_nodes))
    return centers[:n], nlab[:n], nsize[:n], pnode


def graph_edges(points, point_node, centers):
    """sgpr_graph_edges: the pairwise "closest points near the midpoint" distances of gen_graphs -> float64 [n, n]."""
    lib = load_library()
    points = points.detach().to(torch.float32).contiguous()
    p, stride = points.shape
    n = centers.shape[0]
    dev = points.device
    out = torch.zeros(n, n, dtype=torch.float64, device=dev)
    if n == 0:
        return out
    ws = torch.empty(n * n * 4, dtype=torch.uint8, device=dev)
    centers = centers.to(device=dev, dtype=torch.float64).contiguous()
    point_node = point_node.to(device=dev, dtype=torch.int32).contiguous()
    with torch.cuda.device(dev):
        _raise_if(lib, lib.sgpr_graph_edges(_ptr(points), stride, _ptr(point_node), p, n, _ptr(centers), _ptr(out), _ptr(ws),
                                            n * n * 4, _stream_of(points)))
    return out
