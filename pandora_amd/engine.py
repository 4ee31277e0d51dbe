"""Thin object layer over the C ABI (include/pandora_amd.h): one Engine = one GPU context holding
the resident stereo pair; DeviceCostVolume = a device-resident [H][W][D] float32 volume."""
import ctypes as C

import weakref

import numpy as np

from . import _lib
from ._lib import PmxError, check


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


class DeviceCostVolume:
    """Handle on a cost volume living in HBM.  ``to_host()`` materialises it lazily."""

    def __init__(self, engine, handle, D, d0):
        self.engine = engine
        self.handle = handle
        self.D = int(D)
        self.d0 = int(d0)

    @property
    def shape(self):
        return (self.engine.H, self.engine.W, self.D)

    def to_host(self):
        out = np.empty(self.shape, np.float32)
        check(_lib.lib().pmx_cv_download(self.engine.ctx, self.handle, _p(out, C.c_float)), "pmx_cv_download")
        return out

    def rows_to_host(self, row_lo, row_hi):
        """cost_volume[row_lo:row_hi] (float32 [rows][W][D])"""
        H, W, D = self.shape
        out = np.empty((row_hi - row_lo, W, D), np.float32)
        check(_lib.lib().pmx_cv_download_rows(self.engine.ctx, self.handle, int(row_lo), int(row_hi), _p(out, C.c_float)), "pmx_cv_download_rows")
        return out

    def from_host(self, arr):
        arr = np.ascontiguousarray(arr, np.float32)
        if arr.shape != self.shape:
            raise ValueError(f"cost volume shape {arr.shape} != {self.shape}")
        check(_lib.lib().pmx_cv_upload(self.engine.ctx, self.handle, _p(arr, C.c_float)), "pmx_cv_upload")

    def free(self):
        if self.handle is not None and self.engine.ctx is not None:
            _lib.lib().pmx_cv_free(self.engine.ctx, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # interpreter shutdown
            pass


class _PinnedBlock:
    """A page-locked host buffer that numpy arrays are views of; returns to the pool when its last view dies."""
    _pool = {}  # bytes -> [address]: process-wide, pinned memory does not belong to a context
    _pool_bytes = [0]
    POOL_MAX = 2 << 30  # hipHostMalloc / hipHostFree of a 16 MB block cost 2 - 3 ms each: blocks are recycled, up to this many bytes

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        free = self._pool.get(self.nbytes)
        if free:
            self.addr = free.pop()
            self._pool_bytes[0] -= self.nbytes
        else:
            self.addr = _lib.lib().pmx_host_alloc(self.nbytes)
        if not self.addr:
            self.addr = None  # (nothing for __del__ to put into the pool)
            raise MemoryError(f"pmx_host_alloc({nbytes})")
        self.__array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.addr, False), "version": 3}

    def __del__(self):
        try:
            if not getattr(self, "addr", None):
                return  # the allocation failed: there is no block
            if self._pool_bytes[0] + self.nbytes <= self.POOL_MAX:
                self._pool.setdefault(self.nbytes, []).append(self.addr)
                self._pool_bytes[0] += self.nbytes
            else:
                _lib.lib().pmx_host_free(self.addr)
        except Exception:  # interpreter shutdown
            pass


def pinned_empty(shape, dtype):
    """np.empty in page-locked memory (recycled through a small pool): the destination of result downloads."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    try:
        return np.asarray(_PinnedBlock(max(n, 1)))[:n].view(dtype).reshape(shape)
    except MemoryError:  # no page-locked memory left: a pageable destination is slower, not wrong
        return np.empty(shape, dtype)


class DeviceMapArray:
    """A 2-D result of the engine (disparity, validity mask, interpolated coefficient) that has not left the GPU: ``.data``
    downloads it on first use and keeps the host copy - from then on the host copy is the truth, as for any DataArray (callers
    edit ``.data`` in place).  A step that consumes the maps right away (refinement after WTA) never pays the round trip."""

    def __init__(self, engine, which, dims=("row", "col"), coords=None):
        self.engine, self.which = engine, which
        self.dims, self.coords = tuple(dims), dict(coords or {})
        self._shape = (engine.H, engine.W)
        self._host = None
        self._snap = None  # device-side copy taken when the engine's maps moved on before anybody read this one
        self._token = engine.maps_token
        engine._lazy_maps.add(self)

    def on_device(self):
        """True while nobody has looked at (or replaced) the values and the engine still holds exactly them."""
        return self._host is None and self._snap is None and self._token is self.engine.maps_token

    @classmethod
    def from_snapshot(cls, engine, which, snap, coords=None, dims=("row", "col")):
        """A map that lives in a device-side snapshot of its own from the start (the output of a step on snapshots)."""
        self = cls.__new__(cls)
        self.engine, self.which = engine, which
        self.dims, self.coords = tuple(dims), dict(coords or {})
        self._shape = (engine.H, engine.W)
        self._host, self._snap, self._token = None, snap, None
        engine._snapshots.add(self)
        return self

    def device_snapshot(self):
        """The snapshot handle that holds these values on the device - taking it out of the engine's current maps if that is
        where they still are - or None when the host copy is the truth (somebody read or assigned ``.data``)."""
        if self._host is not None:
            return None
        if self._snap is None:
            if self._token is not self.engine.maps_token:
                raise RuntimeError("a device-resident map outlived its values (engine bookkeeping error)")
            self.detach()
        return self._snap

    def detach(self):
        """The engine's maps are about to change: keep these values in a device-side copy of their own (nothing waits, nothing
        crosses PCIe unless somebody reads them)."""
        if self._host is None and self._snap is None:
            self._snap = self.engine.snapshot(self.which)
            self.engine._snapshots.add(self)

    def __del__(self):
        try:
            if self._snap is not None:
                self.engine.free_snapshot(self._snap)
                self._snap = None
        except Exception:  # interpreter shutdown, engine already closed
            pass

    def rebind(self):
        """The engine's current maps are this variable's new value (the step that produced them updated it in place)."""
        self._host = None
        self._drop_snapshot()
        self._token = self.engine.maps_token
        self.engine._lazy_maps.add(self)

    def _drop_snapshot(self):
        if self._snap is not None:
            self.engine.free_snapshot(self._snap)
            self._snap = None

    @property
    def data(self):
        if self._host is None:
            if self._snap is not None:
                self._host = self.engine.read_snapshot(self._snap, self.which, self._shape)
                self._drop_snapshot()
            elif self._token is not self.engine.maps_token:
                raise RuntimeError("a device-resident map outlived its values (engine bookkeeping error)")
            else:
                self._host = self.engine.fetch_map(self.which)
        return self._host

    @data.setter
    def data(self, value):
        self._host = value
        self._drop_snapshot()

    values = data

    @property
    def shape(self):
        return self._shape if self._host is None else self._host.shape

    def sel(self, indexers=None, **kw):
        from .dataset import DataArray

        return DataArray(self.data, self.dims, self.coords).sel(indexers, **kw)

    def copy(self, deep=True):
        from .dataset import DataArray

        return DataArray(np.array(self.data, copy=True) if deep else self.data, self.dims, dict(self.coords))


class Engine:
    """One MI355X context.  Raises if the HIP library is not built or no GPU is visible."""

    def __init__(self, device=0):
        self.ctx = None
        L = _lib.lib()
        n = L.pmx_device_count()
        if n <= 0:
            raise PmxError("pandora_amd: no HIP device visible (there is no CPU fallback)")
        self.ctx = L.pmx_create(int(device))
        if not self.ctx:
            raise PmxError("pmx_create failed: " + L.pmx_last_error().decode())
        self.device = int(device)
        self.H = self.W = 0
        self.subpix = 1
        self.lazy = True  # library default (pmx_set_lazy)
        # identity of what the device-resident result maps currently hold; DeviceMapArrays of an older token that nobody has
        # read yet are downloaded before the maps change (new_maps), unless the step declares them superseded
        self.maps_token = object()
        self._lazy_maps = weakref.WeakSet()
        self._snapshots = weakref.WeakSet()  # DeviceMapArrays living in a device-side copy of their own

    def close(self):
        if self.ctx:
            try:  # results nobody has read yet must survive the context
                for lazy in list(self._lazy_maps) + list(self._snapshots):
                    if lazy._host is None and (lazy._snap is not None or lazy._token is self.maps_token):
                        lazy.data  # noqa: B018  (downloads)
            except Exception:
                pass
            _lib.lib().pmx_destroy(self.ctx)
        self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- residency ---------------------------------------------------------------------------
    def set_images(self, left, right, subpix=1):
        """Upload the pair; returns the content fingerprints (pmx_host_fingerprint) of the two float32 arrays, taken in the same
        pass over them."""
        left = np.ascontiguousarray(left, np.float32)
        right = np.ascontiguousarray(right, np.float32)
        if left.ndim != 2 or left.shape != right.shape:
            raise ValueError("left/right must be 2-D arrays of the same shape")
        self.new_maps()  # result maps of the previous pair that nobody has read yet keep their values
        self.H, self.W = left.shape
        self.subpix = int(subpix)
        fl, fr = C.c_uint64(), C.c_uint64()
        check(_lib.lib().pmx_set_images_fingerprinted(self.ctx, _p(left, C.c_float), _p(right, C.c_float), self.H, self.W, self.subpix,
                                                      C.byref(fl), C.byref(fr)), "pmx_set_images")
        return fl.value, fr.value  # pmx_host_fingerprint of the two arrays as uploaded

    def swap_images(self):
        """Left and right of the resident pair exchanged on the device (pmx_swap_images)."""
        self.new_maps()
        check(_lib.lib().pmx_swap_images(self.ctx), "pmx_swap_images")

    def set_shifted_right(self, k, img):
        """the k-th shifted right image resampled on the host (spline_order > 1): float32 (H, W - 1)"""
        a = np.ascontiguousarray(img, np.float32)
        if a.shape != (self.H, self.W - 1):
            raise ValueError(f"shifted right image {a.shape} != {(self.H, self.W - 1)}")
        check(_lib.lib().pmx_set_shifted_right(self.ctx, int(k), _p(a, C.c_float)), "pmx_set_shifted_right")

    def _check_map_shape(self, what, arr):
        # the C ABI takes a pointer and reads H x W elements behind it: a shorter array is a read past its end (a GPU memory fault
        # when the runtime pins the caller's pages), so the shape is checked where it is still known
        if arr is not None and tuple(arr.shape) != (self.H, self.W):
            raise ValueError(f"{what}: shape {tuple(arr.shape)} is not the resident pair's {(self.H, self.W)}")

    def set_masks(self, msk_left=None, msk_right=None, valid=0, nodata=1):
        ml = None if msk_left is None else np.ascontiguousarray(msk_left, np.int16)
        mr = None if msk_right is None else np.ascontiguousarray(msk_right, np.int16)
        self._check_map_shape("set_masks (left)", ml)
        self._check_map_shape("set_masks (right)", mr)
        check(_lib.lib().pmx_set_masks(self.ctx, _p(ml, C.c_int16), _p(mr, C.c_int16), int(valid), int(nodata)),
              "pmx_set_masks")

    def set_disparity_grids(self, dmin=None, dmax=None):
        a = None if dmin is None else np.ascontiguousarray(dmin, np.float64)
        b = None if dmax is None else np.ascontiguousarray(dmax, np.float64)
        self._check_map_shape("set_disparity_grids (min)", a)
        self._check_map_shape("set_disparity_grids (max)", b)
        check(_lib.lib().pmx_set_disparity_grids(self.ctx, _p(a, C.c_double), _p(b, C.c_double)), "pmx_set_disparity_grids")

    def set_placement_trials(self, trials):
        """Probe up to `trials` candidates for every new volume-sized buffer and keep the fastest (pmx_set_placement_trials)."""
        check(_lib.lib().pmx_set_placement_trials(self.ctx, int(trials)), "pmx_set_placement_trials")

    def measure_hbm(self, nbytes=4 << 30):
        """GB/s of plain streaming kernels on this device: {"read", "write", "copy"} (pmx_measure_hbm)."""
        r, w, c = C.c_double(0), C.c_double(0), C.c_double(0)
        check(_lib.lib().pmx_measure_hbm(self.ctx, int(nbytes), C.byref(r), C.byref(w), C.byref(c)), "pmx_measure_hbm")
        return {"read": r.value, "write": w.value, "copy": c.value}

    def release_caches(self):
        """Gives the memory the context keeps between calls (freed volumes, the SGM accumulator, the marching kernels' hand-off
        buffer) back to the driver; returns (free, total) bytes of the device afterwards (pmx_release_caches)."""
        f, t = C.c_size_t(0), C.c_size_t(0)
        check(_lib.lib().pmx_release_caches(self.ctx, C.byref(f), C.byref(t)), "pmx_release_caches")
        return f.value, t.value

    def set_lazy(self, on):
        """Lazy exact representations of the volume (default on); off = always float32 (reference-like)."""
        check(_lib.lib().pmx_set_lazy(self.ctx, int(bool(on))), "pmx_set_lazy")
        self.lazy = bool(on)

    def set_option(self, name, value):
        """Forces a kernel route / tuning choice of this context (include/pandora_amd.h pmx_set_option):
        name without the PMX_ prefix, value a short string or None to clear.  The library reads its environment once, when the
        context is created; from then on this is the only way to change a choice."""
        name = name[4:] if name.startswith("PMX_") else name
        check(_lib.lib().pmx_set_option(self.ctx, name.encode(), None if value is None else str(value).encode()), "pmx_set_option")

    def get_option(self, name):
        name = name[4:] if name.startswith("PMX_") else name
        v = _lib.lib().pmx_get_option(self.ctx, name.encode())
        return None if v is None else v.decode()

    @staticmethod
    def option_names():
        out, i = [], 0
        while True:
            n = _lib.lib().pmx_option_name(i)
            if n is None:
                return out
            out.append(n.decode())
            i += 1

    def options_from_env(self):
        """Mirrors the PMX_<name> variables of os.environ as they are NOW into this context's options (set or cleared): for scripts
        that change a route between steps of one process (tools/fuzz_*.py).  The library itself reads the environment only in
        pmx_create."""
        import os

        for n in self.option_names():
            self.set_option(n, os.environ.get("PMX_" + n))

    def alloc_cv(self, D, d0):
        h = _lib.lib().pmx_cv_alloc(self.ctx, int(D), int(d0))
        if not h:
            raise PmxError("pmx_cv_alloc failed: " + _lib.lib().pmx_last_error().decode())
        return DeviceCostVolume(self, h, D, d0)

    # -- steps -------------------------------------------------------------------------------
    def census(self, cv, win):
        check(_lib.lib().pmx_census(self.ctx, cv.handle, int(win)), "pmx_census")

    def sad_ssd(self, cv, win, squared):
        check(_lib.lib().pmx_sad_ssd(self.ctx, cv.handle, int(win), int(bool(squared))), "pmx_sad_ssd")

    def zncc(self, cv, win):
        check(_lib.lib().pmx_zncc(self.ctx, cv.handle, int(win)), "pmx_zncc")

    def cv_masked(self, cv, win):
        check(_lib.lib().pmx_cv_masked(self.ctx, cv.handle, int(win)), "pmx_cv_masked")

    def scale_pixels(self, cv, weights):
        """cost(p, d) *= weights[p] (SGM's use_confidence, plugin_libsgm.rst:38-47); NaN weights leave the pixel alone."""
        w = np.ascontiguousarray(weights, np.float32)
        if w.shape != (self.H, self.W):
            raise ValueError("scale_pixels: one weight per pixel")
        check(_lib.lib().pmx_cv_scale_pixels(self.ctx, cv.handle, _p(w, C.c_float)), "pmx_cv_scale_pixels")

    def nan_pixels(self, cv):
        out = np.empty((self.H, self.W), np.uint8)
        check(_lib.lib().pmx_nan_pixels(self.ctx, cv.handle, _p(out, C.c_uint8)), "pmx_nan_pixels")
        return out.astype(bool)

    def mark_missing(self, cv):
        """Snapshot, kept on the device with the volume, of the pixels whose cost is NaN for every disparity NOW (pmx_cv_mark_missing)."""
        check(_lib.lib().pmx_cv_mark_missing(self.ctx, cv.handle), "pmx_cv_mark_missing")

    def get_missing(self, cv):
        out = np.empty((self.H, self.W), np.uint8)
        check(_lib.lib().pmx_cv_get_missing(self.ctx, cv.handle, _p(out, C.c_uint8)), "pmx_cv_get_missing")
        return out.astype(bool)

    def compose_validity(self, base, missing_of=None, border=0):
        """The validity mask the next WTA starts from, put together on the device (pmx_compose_validity): ``base`` int64 (W,) - one
        line for every row - or (H, W); ``missing_of``: volume whose mark_missing snapshot sets the missing-range bit; ``border``:
        width of the frame that becomes PANDORA_MSK_PIXEL_LEFT_NODATA_OR_BORDER."""
        self.new_maps()
        b = np.ascontiguousarray(base, np.int64)
        if b.shape not in ((self.W,), (self.H, self.W)):
            raise ValueError(f"compose_validity: base {b.shape} is neither ({self.W},) nor ({self.H}, {self.W})")
        check(_lib.lib().pmx_compose_validity(self.ctx, _p(b, C.c_int64), 1 if b.ndim == 1 else self.H,
                                              None if missing_of is None else missing_of.handle, int(border)), "pmx_compose_validity")

    def reverse_cost_volume(self, cv, min_disp):
        h = _lib.lib().pmx_reverse_cost_volume(self.ctx, cv.handle, int(min_disp))
        if not h:
            raise PmxError("pmx_reverse_cost_volume failed: " + _lib.lib().pmx_last_error().decode())
        return DeviceCostVolume(self, h, cv.D, min_disp)

    def cbca(self, cv, offset, intensity, distance):
        check(_lib.lib().pmx_cbca(self.ctx, cv.handle, int(offset), float(intensity), int(distance)), "pmx_cbca")

    def cross_support(self, side, offset, intensity, distance):
        wd = self.W if side <= 1 else self.W - 1
        out = np.empty((self.H - 2 * offset, wd - 2 * offset, 4), np.int16)
        check(_lib.lib().pmx_cross_support(self.ctx, int(side), int(offset), float(intensity), int(distance),
                                           _p(out, C.c_int16)), "pmx_cross_support")
        return out

    def sgm(self, cv, P1, P2, is_max=False, invalid_cost=0.0, overcounting=False, dir_mask=0xFF):
        """dir_mask: test hook (pmx_debug_sgm_directions), bit k = k-th path of the definition's order; 0xff = all eight."""
        if dir_mask != 0xFF:
            check(_lib.lib().pmx_debug_sgm_directions(self.ctx, int(dir_mask)), "pmx_debug_sgm_directions")
        try:
            check(_lib.lib().pmx_sgm(self.ctx, cv.handle, float(P1), float(P2), int(bool(is_max)), float(invalid_cost),
                                     int(bool(overcounting))), "pmx_sgm")
        finally:
            if dir_mask != 0xFF:
                check(_lib.lib().pmx_debug_sgm_directions(self.ctx, 0xFF), "pmx_debug_sgm_directions")

    def sgm_p2maps(self, cv, P1, p2maps, is_max=False, invalid_cost=0.0, overcounting=False):
        """pmx_sgm_p2maps: P2 per pixel and path direction, float32 [8][H][W] in the definition's order of the directions."""
        maps = np.ascontiguousarray(p2maps, np.float32)
        if maps.shape != (8, self.H, self.W):
            raise ValueError(f"sgm_p2maps: the penalty maps must be [8][{self.H}][{self.W}], got {maps.shape}")
        check(_lib.lib().pmx_sgm_p2maps(self.ctx, cv.handle, float(P1), _p(maps, C.c_float), int(bool(is_max)), float(invalid_cost),
                                        int(bool(overcounting))), "pmx_sgm_p2maps")

    def new_maps(self, superseded=()):
        """Call BEFORE an operation that overwrites the device-resident result maps: pending DeviceMapArrays move into a
        device-side copy of their own (DeviceMapArray.detach) unless they are in `superseded` (the operation updates exactly those
        variables in place)."""
        for lazy in list(self._lazy_maps):
            if lazy._host is None and lazy._token is self.maps_token and not any(lazy is s for s in superseded):
                lazy.detach()
        self._lazy_maps.clear()
        self.maps_token = object()

    def snapshot(self, which):
        h = _lib.lib().pmx_map_snapshot(self.ctx, {"disp": 0, "validity": 1, "itp": 2}[which])
        if not h:
            raise PmxError("pmx_map_snapshot failed: " + _lib.lib().pmx_last_error().decode())
        return h

    def alloc_snapshot(self, which):
        """An uninitialised snapshot (float32 for "disp" / "itp" / "conf", int64 for "validity")."""
        h = _lib.lib().pmx_map_snapshot_alloc(self.ctx, 1 if which == "validity" else 0)
        if not h:
            raise PmxError("pmx_map_snapshot_alloc failed: " + _lib.lib().pmx_last_error().decode())
        return h

    def maps_restore(self, disp_snap, validity_snap):
        """The engine's disparity / validity maps become what the two snapshots hold (pmx_maps_restore)."""
        self.new_maps()
        check(_lib.lib().pmx_maps_restore(self.ctx, disp_snap, validity_snap), "pmx_maps_restore")

    def median_filter_maps(self, disp_snap, validity_snap, filter_size):
        out = self.alloc_snapshot("disp")
        try:
            check(_lib.lib().pmx_median_filter_maps(self.ctx, disp_snap, validity_snap, int(filter_size), out), "pmx_median_filter_maps")
        except Exception:
            self.free_snapshot(out)
            raise
        return out

    def cross_checking_maps(self, disp_left, validity_left, disp_right, dmin, dmax, threshold, border=0):
        """validation.py:226-371 on snapshots: ``validity_left`` is updated in place (then framed, criteria.mask_border, when
        ``border`` > 0); returns the snapshot of the left-right distance."""
        conf = self.alloc_snapshot("conf")
        try:
            check(_lib.lib().pmx_cross_checking_maps(self.ctx, disp_left, validity_left, disp_right, int(dmin), int(dmax),
                                                     float(threshold), conf), "pmx_cross_checking_maps")
            if border > 0:
                check(_lib.lib().pmx_validity_frame_map(self.ctx, validity_left, int(border)), "pmx_validity_frame_map")
        except Exception:
            self.free_snapshot(conf)
            raise
        return conf

    def read_snapshot(self, snap, which, shape):
        out = pinned_empty(shape, np.int64 if which == "validity" else np.float32)
        check(_lib.lib().pmx_map_snapshot_read(self.ctx, snap, out.ctypes.data), "pmx_map_snapshot_read")
        return out

    def free_snapshot(self, snap):
        if self.ctx:
            _lib.lib().pmx_map_snapshot_free(self.ctx, snap)

    def fetch_map(self, which):
        """One of the device-resident result maps into pinned host memory."""
        out = pinned_empty((self.H, self.W), np.int64 if which == "validity" else np.float32)
        ptrs = {"disp": (out, None, None), "validity": (None, out, None), "itp": (None, None, out)}[which]
        check(_lib.lib().pmx_get_disparity(self.ctx, _p(ptrs[0], C.c_float), _p(ptrs[1], C.c_int64), _p(ptrs[2], C.c_float)),
              "pmx_get_disparity")
        return out

    def set_validity(self, validity=None):
        self.new_maps()
        v = None if validity is None else np.ascontiguousarray(validity, np.int64)
        self._check_map_shape("set_validity", v)
        check(_lib.lib().pmx_set_validity(self.ctx, _p(v, C.c_int64)), "pmx_set_validity")

    def wta(self, cv, is_max=False, invalid_disparity=-9999.0):
        self.new_maps()
        check(_lib.lib().pmx_wta(self.ctx, cv.handle, int(bool(is_max)), float(invalid_disparity)), "pmx_wta")

    def refine(self, cv, method, is_max=False, superseded=()):
        self.new_maps(superseded)
        m = {"vfit": 0, "quadratic": 1}[method]
        check(_lib.lib().pmx_refine(self.ctx, cv.handle, m, int(bool(is_max))), "pmx_refine")

    def refine_approximate(self, cv_left, method, is_max=False):
        """pmx_refine_approximate: the resident map as a RIGHT map refined on the left volume's diagonals
        (refinement_cpp.loop_approximate_refinement)."""
        self.new_maps()
        m = {"vfit": 0, "quadratic": 1}[method]
        check(_lib.lib().pmx_refine_approximate(self.ctx, cv_left.handle, m, int(bool(is_max))), "pmx_refine_approximate")

    def get_disparity(self, want_itp=False, out=None):
        """Download disp float32, validity int64 (and interpolated_coeff).  `out` = (disp, validity[, itp]) arrays of
        the right shape/dtype to fill in place: a caller streaming pairs of one shape avoids 67 MB of first-touch
        page faults per pair at 2048x2048."""
        if out is not None:
            disp, val = out[0], out[1]
            itp = out[2] if want_itp else None
            for a, dt in ((disp, np.float32), (val, np.int64)) + (((itp, np.float32),) if want_itp else ()):
                if a.shape != (self.H, self.W) or a.dtype != dt or not a.flags["C_CONTIGUOUS"]:
                    raise ValueError("get_disparity: `out` arrays must be C-contiguous (H, W) float32 / int64 / float32")
        else:  # page-locked, pooled: no first-touch page faults under the DMA
            disp = pinned_empty((self.H, self.W), np.float32)
            val = pinned_empty((self.H, self.W), np.int64)
            itp = pinned_empty((self.H, self.W), np.float32) if want_itp else None
        check(_lib.lib().pmx_get_disparity(self.ctx, _p(disp, C.c_float), _p(val, C.c_int64), _p(itp, C.c_float)),
              "pmx_get_disparity")
        return (disp, val, itp) if want_itp else (disp, val)

    def set_disparity(self, disp=None, validity=None):
        self.new_maps()
        d = None if disp is None else np.ascontiguousarray(disp, np.float32)
        v = None if validity is None else np.ascontiguousarray(validity, np.int64)
        self._check_map_shape("set_disparity (disparity map)", d)
        self._check_map_shape("set_disparity (validity mask)", v)
        check(_lib.lib().pmx_set_disparity(self.ctx, _p(d, C.c_float), _p(v, C.c_int64)), "pmx_set_disparity")

    def wta_minkey(self, cv, is_max, index_offset, dev_keys_ptr):
        check(_lib.lib().pmx_wta_minkey(self.ctx, cv.handle, int(bool(is_max)), int(index_offset), C.c_void_p(dev_keys_ptr)),
              "pmx_wta_minkey")

    def wta_from_keys(self, dev_keys_ptr, d0_global, subpix, invalid_disparity):
        self.new_maps()
        check(_lib.lib().pmx_wta_from_keys(self.ctx, C.c_void_p(dev_keys_ptr), float(d0_global), int(subpix),
                                           float(invalid_disparity)), "pmx_wta_from_keys")

    # -- validation -------------------------------------------------------------
    def cross_checking(self, disp_left, validity_left, disp_right, dmin, dmax, threshold):
        """validation.py:226-371 on the device; returns (validity int64 updated copy, confidence float32)."""
        dl = np.ascontiguousarray(disp_left, np.float32)
        dr = np.ascontiguousarray(disp_right, np.float32)
        if dl.ndim != 2 or dl.shape != dr.shape:
            raise ValueError("cross_checking: left/right disparity maps must be 2-D and of the same shape")
        val = np.array(validity_left, np.int64, order="C", copy=True)
        if val.shape != dl.shape:
            raise ValueError(f"cross_checking: validity mask {val.shape} != disparity maps {dl.shape}")
        conf = np.empty(dl.shape, np.float32)
        check(_lib.lib().pmx_cross_checking(self.ctx, _p(dl, C.c_float), _p(val, C.c_int64), _p(dr, C.c_float), dl.shape[0],
                                            dl.shape[1], int(dmin), int(dmax), float(threshold), _p(conf, C.c_float)),
              "pmx_cross_checking")
        return val, conf

    def reverse_disp_range(self, left_min, left_max):
        """matching_cost.cpp:59-132 on the device; returns (right_min, right_max) float32."""
        a = np.ascontiguousarray(left_min, np.float32)
        b = np.ascontiguousarray(left_max, np.float32)
        if a.ndim != 2 or a.shape != b.shape:
            raise ValueError("reverse_disp_range: min/max grids must be 2-D and of the same shape")
        if np.all(np.isnan(a)) or np.all(np.isnan(b)):
            return np.full(a.shape, np.nan, np.float32), np.full(a.shape, np.nan, np.float32)
        gmin, gmax = int(np.nanmin(a)), int(np.nanmax(b))
        rmin, rmax = np.empty(a.shape, np.float32), np.empty(a.shape, np.float32)
        check(_lib.lib().pmx_reverse_disp_range(self.ctx, _p(a, C.c_float), _p(b, C.c_float), a.shape[0], a.shape[1], gmin, gmax,
                                                _p(rmin, C.c_float), _p(rmax, C.c_float)), "pmx_reverse_disp_range")
        return rmin, rmax

    INTERPOLATION_PASSES = {"occlusion_mc_cnn": 0, "mismatch_mc_cnn": 1, "occlusion_sgm": 2, "mismatch_sgm": 3}

    def interpolate_disparity(self, disp, validity, passes):
        """interpolated_disparity.cpp on the device; ``passes``: names of INTERPOLATION_PASSES, run in order.
        Returns (disparity float32, validity int64), inputs untouched."""
        d = np.array(disp, np.float32, order="C", copy=True)
        v = np.array(validity, np.int64, order="C", copy=True)
        if d.ndim != 2 or d.shape != v.shape:
            raise ValueError("interpolate_disparity: disparity map and validity mask must be 2-D and of the same shape")
        codes = (C.c_int * len(passes))(*[self.INTERPOLATION_PASSES[p] for p in passes])
        check(_lib.lib().pmx_interpolate_disparity(self.ctx, _p(d, C.c_float), _p(v, C.c_int64), d.shape[0], d.shape[1], codes,
                                                   len(passes)), "pmx_interpolate_disparity")
        return d, v

    # -- disparity filters -------------------------------------------------------
    def median_filter_disparity(self, disp, validity, filter_size):
        """median.py:94-179 on the device; returns the filtered float32 map (input untouched)."""
        d = np.array(disp, np.float32, order="C", copy=True)
        v = np.ascontiguousarray(validity, np.int64)
        if d.ndim != 2 or d.shape != v.shape:
            raise ValueError("median_filter_disparity: disparity map and validity mask must be 2-D and of the same shape")
        check(_lib.lib().pmx_median_filter_disparity(self.ctx, _p(d, C.c_float), _p(v, C.c_int64), d.shape[0], d.shape[1],
                                                     int(filter_size)), "pmx_median_filter_disparity")
        return d

    def denoise_disparity(self, disp, validity, color, grad_row, grad_col, filter_size, sigma_euclidian, sigma_color, sigma_planar):
        """disparity_denoiser.py:223-313 on the device (grad_* = np.gradient of the blurred map); returns the filtered float32 map."""
        d = np.array(disp, dtype=np.float32, order="C", copy=True)
        v = np.ascontiguousarray(validity, np.int64)
        maps = [np.ascontiguousarray(m, np.float32) for m in (color, grad_row, grad_col)]
        if d.ndim != 2 or any(m.shape != d.shape for m in maps) or v.shape != d.shape:
            raise ValueError("denoise_disparity: disparity map, validity mask, colour band and gradients must be 2-D and of one shape")
        check(_lib.lib().pmx_denoise_disparity(self.ctx, _p(d, C.c_float), _p(v, C.c_int64), _p(maps[0], C.c_float), _p(maps[1], C.c_float),
                                               _p(maps[2], C.c_float), d.shape[0], d.shape[1], int(filter_size), float(sigma_euclidian),
                                               float(sigma_color), float(sigma_planar)), "pmx_denoise_disparity")
        return d

    def bilateral_filter_disparity(self, disp, validity, sigma_color, sigma_space):
        """bilateral.py:100-255 on the device; returns the filtered float32 map (input untouched)."""
        d = np.array(disp, np.float32, order="C", copy=True)
        v = np.ascontiguousarray(validity, np.int64)
        if d.ndim != 2 or d.shape != v.shape:
            raise ValueError("bilateral_filter_disparity: disparity map and validity mask must be 2-D and of the same shape")
        check(_lib.lib().pmx_bilateral_filter_disparity(self.ctx, _p(d, C.c_float), _p(v, C.c_int64), d.shape[0], d.shape[1],
                                                        float(sigma_color), float(sigma_space)), "pmx_bilateral_filter_disparity")
        return d

    # -- multiscale ----------------------------------------------------------------
    def interpolate_nodata(self, img, msk, invalid_bits, filled_value):
        """img_tools.cpp:99-155 (interpolate_nodata_sgm) on the device -> (filled float32 image, int32 mask)."""
        im = np.ascontiguousarray(img, np.float32)
        mk = np.ascontiguousarray(msk, np.int32)
        if im.ndim != 2 or im.shape != mk.shape:
            raise ValueError("interpolate_nodata: image and mask must be 2-D and of the same shape")
        out_i, out_m = np.empty_like(im), np.empty_like(mk)
        check(_lib.lib().pmx_interpolate_nodata(self.ctx, _p(im, C.c_float), _p(mk, C.c_int32), im.shape[0], im.shape[1],
                                                int(invalid_bits), int(filled_value), _p(out_i, C.c_float), _p(out_m, C.c_int32)),
              "pmx_interpolate_nodata")
        return out_i, out_m

    def disparity_range(self, disp, validity, window_size, marge, global_min, global_max):
        """fixed_zoom_pyramid.py:106-172 (before the zoom) on the device -> (range_min, range_max) float32."""
        d = np.ascontiguousarray(disp, np.float32)
        v = np.ascontiguousarray(validity, np.int64)
        if d.ndim != 2 or d.shape != v.shape:
            raise ValueError("disparity_range: disparity map and validity mask must be 2-D and of the same shape")
        lo, hi = np.empty(d.shape, np.float32), np.empty(d.shape, np.float32)
        check(_lib.lib().pmx_disparity_range(self.ctx, _p(d, C.c_float), _p(v, C.c_int64), d.shape[0], d.shape[1], int(window_size),
                                             int(marge), int(global_min), int(global_max), _p(lo, C.c_float), _p(hi, C.c_float)),
              "pmx_disparity_range")
        return lo, hi

    # -- cost-volume confidence -------------------------------------------------------
    def order_statistics(self, values, ranks):
        """float32 array of the ranks-th smallest values of ``values`` (0-based, NaNs last): pmx_order_statistics."""
        v = np.ascontiguousarray(values, np.float32).ravel()
        r = np.ascontiguousarray(ranks, np.uint64)
        out = np.empty(len(r), np.float32)
        check(_lib.lib().pmx_order_statistics(self.ctx, _p(v, C.c_float), v.size, r.ctypes.data_as(C.POINTER(C.c_size_t)), len(r),
                                              _p(out, C.c_float)), "pmx_order_statistics")
        return out

    def ambiguity(self, cv, etas, grid_min, grid_max, negate=False):
        """ambiguity.cpp:28-142 on the resident volume -> float32 [H][W] integral of the ambiguity (not normalised)."""
        e = np.ascontiguousarray(etas, np.float32)
        gmin, gmax = self._grids("ambiguity", grid_min, grid_max)
        out = pinned_empty((self.H, self.W), np.float32)  # (page-locked, recycled: no first-touch page faults under the copy)
        check(_lib.lib().pmx_ambiguity(self.ctx, cv.handle, _p(e, C.c_float), len(e), _p(gmin, C.c_int64), _p(gmax, C.c_int64),
                                       int(bool(negate)), _p(out, C.c_float)), "pmx_ambiguity")
        return out

    def _grids(self, what, grid_min, grid_max):
        if grid_min is None and grid_max is None:  # every pixel searches the volume's whole range
            return None, None
        if (grid_min is None) != (grid_max is None):
            raise ValueError(f"{what}: give both disparity grids or neither")
        gmin = np.ascontiguousarray(grid_min, np.int64)
        gmax = np.ascontiguousarray(grid_max, np.int64)
        if gmin.shape != (self.H, self.W) or gmax.shape != (self.H, self.W):
            raise ValueError(f"{what}: the disparity grids must have the image shape")
        return gmin, gmax

    def risk(self, cv, etas, grid_min, grid_max, negate=False):
        """risk.cpp:28-197 as risk.py:144-166 drives it, on the resident volume -> (risk_max, risk_min, disp_sup, disp_inf)."""
        e = np.ascontiguousarray(etas, np.float64)
        gmin, gmax = self._grids("risk", grid_min, grid_max)
        outs = [pinned_empty((self.H, self.W), np.float32) for _ in range(4)]
        check(_lib.lib().pmx_risk(self.ctx, cv.handle, _p(e, C.c_double), len(e), _p(gmin, C.c_int64), _p(gmax, C.c_int64),
                                  int(bool(negate)), *[_p(o, C.c_float) for o in outs]), "pmx_risk")
        return tuple(outs)

    def interval_bounds(self, cv, possibility_threshold, type_factor, grid_min, grid_max):
        """interval_bounds.cpp:28-161 on the resident volume -> (interval_inf, interval_sup) float32 [H][W]."""
        gmin, gmax = self._grids("interval_bounds", grid_min, grid_max)
        lo, hi = pinned_empty((self.H, self.W), np.float32), pinned_empty((self.H, self.W), np.float32)
        check(_lib.lib().pmx_interval_bounds(self.ctx, cv.handle, float(possibility_threshold), float(type_factor), _p(gmin, C.c_int64),
                                             _p(gmax, C.c_int64), _p(lo, C.c_float), _p(hi, C.c_float)), "pmx_interval_bounds")
        return lo, hi

    def debug_small_division(self):
        """how many of the 65536 x 1024 small-integer quotients of the census + CBCA marching kernel differ from the IEEE division"""
        bad = C.c_uint(0)
        check(_lib.lib().pmx_debug_small_division(self.ctx, C.byref(bad)), "pmx_debug_small_division")
        return bad.value

    def debug_path_costs(self, cv, raw=False):
        """uint8 [8][H][W][D] per-direction SGM path costs of a volume in the fused representation
        (raw=True: the device byte order [8][H][W][Dp] and the (gl, kpl) lane map)."""
        dp, gl, kpl = C.c_int(0), C.c_int(0), C.c_int(0)
        fn = _lib.lib().pmx_debug_path_costs
        check(fn(self.ctx, cv.handle, None, 0, C.byref(dp), C.byref(gl), C.byref(kpl)), "pmx_debug_path_costs")
        out = np.empty((8, self.H, self.W, dp.value), np.uint8)
        check(fn(self.ctx, cv.handle, _p(out, C.c_uint8), out.nbytes, C.byref(dp), C.byref(gl), C.byref(kpl)), "pmx_debug_path_costs")
        if raw:
            return out, gl.value, kpl.value
        d = np.arange(cv.D)
        s, k, m4 = d // kpl.value, d % kpl.value, kpl.value & ~3
        nact = -(-cv.D // kpl.value)
        pos = np.where(k < m4, s * m4 + k, nact * m4 + s)
        return out[:, :, :, pos]

    # -- one pair over several GPUs (csrc/pmx_comm.hip; pandora_amd.comm.Comm bootstraps it) ----------------------------
    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        check(_lib.lib().pmx_comm_unique_id(buf, 128), "pmx_comm_unique_id")
        return buf.raw

    def comm_init(self, uid, world, rank):
        check(_lib.lib().pmx_comm_init(self.ctx, C.c_char_p(bytes(uid)), len(uid), int(world), int(rank)), "pmx_comm_init")

    def comm_count(self):
        n = C.c_int(0)
        check(_lib.lib().pmx_comm_count(self.ctx, C.byref(n)), "pmx_comm_count")
        return n.value

    def comm_destroy(self):
        check(_lib.lib().pmx_comm_destroy(self.ctx), "pmx_comm_destroy")

    def comm_allreduce(self, which, op):
        from .comm import OPS

        check(_lib.lib().pmx_comm_allreduce(self.ctx, _lib.XBUFS[which][0], OPS[op]), "pmx_comm_allreduce")

    def comm_allreduce_scalars(self, eight, op):
        from .comm import OPS

        a = np.ascontiguousarray(eight, np.float64).copy()
        assert a.size == 8
        check(_lib.lib().pmx_comm_allreduce_scalars(self.ctx, _p(a, C.c_double), OPS[op]), "pmx_comm_allreduce_scalars")
        return a

    def comm_allgather_rows(self, with_itp):
        check(_lib.lib().pmx_comm_allgather_rows(self.ctx, int(bool(with_itp))), "pmx_comm_allgather_rows")

    def comm_gather_rows(self, root, with_itp):
        check(_lib.lib().pmx_comm_gather_rows(self.ctx, int(root), int(bool(with_itp))), "pmx_comm_gather_rows")

    def xbuf_download(self, which):
        """TEST TRANSPORT ONLY (two ranks on one GPU): host copy of an exchange buffer."""
        idx, dtype = _lib.XBUFS[which]
        n = C.c_size_t(0)
        check(_lib.lib().pmx_xbuf_info(self.ctx, idx, C.byref(n), None), "pmx_xbuf_info")
        out = np.empty(n.value, dtype)
        check(_lib.lib().pmx_xbuf_download(self.ctx, idx, out.ctypes.data_as(C.c_void_p)), "pmx_xbuf_download")
        return out

    def xbuf_upload(self, which, arr):
        idx, dtype = _lib.XBUFS[which]
        a = np.ascontiguousarray(arr, dtype).ravel()
        n = C.c_size_t(0)
        check(_lib.lib().pmx_xbuf_info(self.ctx, idx, C.byref(n), None), "pmx_xbuf_info")
        if a.size != n.value:
            raise ValueError(f"xbuf_upload: exchange buffer {which!r} holds {n.value} elements, got {a.size}")
        check(_lib.lib().pmx_xbuf_upload(self.ctx, idx, a.ctypes.data_as(C.c_void_p)), "pmx_xbuf_upload")

    def shard_minkey(self, cv, is_max, index_offset):
        check(_lib.lib().pmx_shard_minkey(self.ctx, cv.handle, int(bool(is_max)), int(index_offset)), "pmx_shard_minkey")

    def shard_from_keys(self, d0_global, subpix, invalid_disparity):
        self.new_maps()
        check(_lib.lib().pmx_shard_from_keys(self.ctx, float(d0_global), int(subpix), float(invalid_disparity)), "pmx_shard_from_keys")

    def shard_nan_pixels(self, cv):
        check(_lib.lib().pmx_shard_nan_pixels(self.ctx, cv.handle), "pmx_shard_nan_pixels")

    def shard_refine_pack(self, cv, method, is_max, own_lo, own_hi, last_rank):
        self.new_maps()
        check(_lib.lib().pmx_shard_refine_pack(self.ctx, cv.handle, {"vfit": 0, "quadratic": 1}[method], int(bool(is_max)), float(own_lo), float(own_hi),
                                               int(bool(last_rank))), "pmx_shard_refine_pack")

    def shard_refine_unpack(self):
        self.new_maps()
        check(_lib.lib().pmx_shard_refine_unpack(self.ctx), "pmx_shard_refine_unpack")

    def tile_place(self, full_H, own_lo, own_hi, tile_lo, with_itp):
        check(_lib.lib().pmx_tile_place(self.ctx, int(full_H), int(own_lo), int(own_hi), int(tile_lo), int(bool(with_itp))), "pmx_tile_place")

    def set_full_rows(self, full_H, own_lo, own_hi, disp, validity, itp=None):
        d = np.ascontiguousarray(disp, np.float32)
        v = np.ascontiguousarray(validity, np.int64)
        t = None if itp is None else np.ascontiguousarray(itp, np.float32)
        rows = (int(own_hi) - int(own_lo), self.W)
        if d.shape != rows or v.shape != rows or (t is not None and t.shape != rows):
            raise ValueError(f"set_full_rows: rows [{own_lo}, {own_hi}) are {rows} per map, got {d.shape} / {v.shape}"
                             + ("" if t is None else f" / {t.shape}"))
        check(_lib.lib().pmx_set_full_rows(self.ctx, int(full_H), int(own_lo), int(own_hi), d.ctypes.data_as(C.c_void_p),
                                           v.ctypes.data_as(C.c_void_p), None if t is None else t.ctypes.data_as(C.c_void_p)), "pmx_set_full_rows")

    def get_full_maps(self, full_H, want_itp=False):
        n = C.c_size_t(0)
        check(_lib.lib().pmx_xbuf_info(self.ctx, _lib.XBUFS["full_disp"][0], C.byref(n), None), "pmx_xbuf_info")
        if n.value != int(full_H) * self.W:  # the library copies what tile_place / set_full_rows sized, not what the caller expects
            raise ValueError(f"get_full_maps: the placed maps hold {n.value} pixels, not {full_H} x {self.W}")
        disp = np.empty((full_H, self.W), np.float32)
        val = np.empty((full_H, self.W), np.int64)
        itp = np.empty((full_H, self.W), np.float32) if want_itp else None
        check(_lib.lib().pmx_get_full_maps(self.ctx, disp.ctypes.data_as(C.c_void_p), val.ctypes.data_as(C.c_void_p),
                                           itp.ctypes.data_as(C.c_void_p) if want_itp else None), "pmx_get_full_maps")
        return (disp, val, itp) if want_itp else (disp, val)

    # -- measurement -------------------------------------------------------------------------
    def sync(self):
        check(_lib.lib().pmx_sync(self.ctx), "pmx_sync")

    def set_profiling(self, on):
        check(_lib.lib().pmx_set_profiling(self.ctx, int(bool(on))), "pmx_set_profiling")

    def reset_stage_times(self):
        check(_lib.lib().pmx_reset_stage_times(self.ctx), "pmx_reset_stage_times")

    def stage_time(self, name):
        ms = C.c_double(0)
        n = C.c_int(0)
        check(_lib.lib().pmx_stage_time(self.ctx, _lib.STAGES[name], C.byref(ms), C.byref(n)), "pmx_stage_time")
        return ms.value, n.value

    def stream(self):
        return _lib.lib().pmx_stream(self.ctx)
