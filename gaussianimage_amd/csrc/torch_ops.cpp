// torch_ops.cpp -- the three operators the reference imports
// (project_gaussians_2d, project_gaussians_2d_scale_rot, rasterize_gaussians_sum:
// /root/reference/gaussianimage_cholesky.py:1-2,77,81-82, /root/reference/gaussianimage_rs.py:1-2,75-77)
// as C++ autograd functions over the gi2d_* C ABI.  Host code only (compiled with g++): torch
// supplies device memory, the current stream and the autograd graph; every launch goes through
// libgi2d.so.  The Python binding (ops.py) has the same behaviour and is what this replaces on
// the hot path: argument checks, per-stream workspaces, the tile lists shared by the raster
// passes of one model forward, the handling of a late overflow report.
#include <torch/extension.h>

#include <c10/hip/HIPStream.h>

#include <climits>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>

#include "gi2d.h"

#ifndef GI2D_TORCH_OPS_ID
#define GI2D_TORCH_OPS_ID "unknown"
#endif

namespace py = pybind11;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

namespace {

const gi2d_conv *g_conv = nullptr;  // the Python side's convention struct (set_conventions edits it)
PyObject *g_error_class = nullptr;  // gaussianimage_amd._lib.Gi2dError

// Identity of the tensors a binning was made from (ops.py: Workspace._key).
struct BinKey {
    const void *xys = nullptr, *radii = nullptr;
    uint32_t xys_version = 0, radii_version = 0;
    int64_t N = -1;
    int H = 0, W = 0;
    bool valid = false;
    bool operator==(const BinKey &o) const
    {
        return valid && o.valid && xys == o.xys && radii == o.radii &&
               xys_version == o.xys_version && radii_version == o.radii_version && N == o.N &&
               H == o.H && W == o.W;
    }
};

BinKey make_key(const at::Tensor &xys, const at::Tensor &radii, int H, int W)
{
    BinKey k;
    k.xys = xys.data_ptr();
    k.radii = radii.data_ptr();
    k.xys_version = xys._version();
    k.radii_version = radii._version();
    k.N = xys.size(0);
    k.H = H;
    k.W = W;
    k.valid = true;
    return k;
}

// The operators' workspace of one (device, stream): the library handle plus the host-side state
// that decides when the lists can be reused.  Used by one stream at a time, like the handle.
struct WsState {
    gi2d_workspace *handle = nullptr;
    int device = 0;
    BinKey key, counted_key;
    at::Tensor keep_xys, keep_radii;  // strong references: their storage must not be recycled
    at::Tensor counted_xys, counted_radii;  // the same for the tensors behind counted_key
    int64_t late_overflows = 0;
    bool strict = false, sync_binning = false;

    WsState() = default;
    WsState(const WsState &) = delete;
    WsState &operator=(const WsState &) = delete;
    ~WsState()
    {
        if (handle != nullptr) gi2d_workspace_destroy(handle);
    }
    uintptr_t handle_value() const { return reinterpret_cast<uintptr_t>(handle); }
    void invalidate(bool counted)
    {
        key.valid = false;
        if (counted) drop_counted();
    }
    void drop_counted()
    {
        counted_key.valid = false;
        counted_xys = at::Tensor();
        counted_radii = at::Tensor();
    }
    // The projection that produced xys / radii also filled the lists.  Detached aliases hold the
    // storage (and no autograd graph): while the key is valid the allocator cannot hand the same
    // addresses, with a fresh version counter, to unrelated tensors.
    void set_counted(const at::Tensor &xys, const at::Tensor &radii, int H, int W)
    {
        counted_key = make_key(xys, radii, H, W);
        counted_xys = xys.detach();
        counted_radii = radii.detach();
    }
};

// (on the heap and never freed: no workspace is destroyed while the process shuts down)
using WsMap = std::map<std::pair<int, uintptr_t>, std::shared_ptr<WsState>>;
std::mutex g_ws_lock;
WsMap &g_ws = *new WsMap();

[[noreturn]] void raise_status(int status, const char *where, bool have_gil)
{
    std::ostringstream msg;
    msg << where << ": " << gi2d_error_string(status) << " (status " << status << ")";
    if (have_gil && g_error_class != nullptr) {
        py::object exc = py::reinterpret_borrow<py::object>(g_error_class)(status, where);
        PyErr_SetObject(g_error_class, exc.ptr());
        throw py::error_already_set();
    }
    throw std::runtime_error(msg.str());
}

inline void check(int status, const char *where, bool have_gil)
{
    if (status != GI2D_OK) raise_status(status, where, have_gil);
}

std::shared_ptr<WsState> workspace(int device, uintptr_t stream)
{
    std::lock_guard<std::mutex> hold(g_ws_lock);
    auto &slot = g_ws[{device, stream}];
    if (!slot) {
        auto st = std::make_shared<WsState>();
        st->device = device;
        const char *sync = std::getenv("GI2D_SYNC");
        st->sync_binning = sync != nullptr && sync[0] == '1';
        const int rc = gi2d_workspace_create(device, &st->handle);
        if (rc != GI2D_OK) {
            g_ws.erase({device, stream});
            raise_status(rc, "gi2d_workspace_create", false);
        }
        slot = st;
    }
    return slot;
}

inline void *raw_stream(const at::Tensor &t)
{
    return c10::hip::getCurrentHIPStream(t.get_device()).stream();
}

inline std::shared_ptr<WsState> workspace_of(const at::Tensor &t, void *stream)
{
    return workspace((int)t.get_device(), reinterpret_cast<uintptr_t>(stream));
}

// Builds (or reuses) the tile lists for these projected Gaussians (ops.py: Workspace.bin).
void bin(WsState &ws, const at::Tensor &xys, const at::Tensor &radii, int H, int W, int flags,
         bool force, void *stream, bool have_gil)
{
    const BinKey key = make_key(xys, radii, H, W);
    if (!force && key == ws.key) return;
    const int tiles_x = (W + GI2D_BLOCK - 1) / GI2D_BLOCK, tiles_y = (H + GI2D_BLOCK - 1) / GI2D_BLOCK;
    if (ws.sync_binning) flags |= GI2D_BIN_SYNC;
    if (ws.counted_key == key) flags |= GI2D_BIN_COUNTED;
    ws.drop_counted();
    ws.key.valid = false;
    int rc = gi2d_bin_tiles(ws.handle, xys.numel() ? xys.data_ptr<float>() : nullptr,
                            radii.numel() ? radii.data_ptr<int>() : nullptr, (int)xys.size(0),
                            tiles_x, tiles_y, flags, stream);
    if (rc == GI2D_E_PREV_OVERFLOW) {
        // an EARLIER asynchronous frame outgrew a tile list: its overflowed tiles were NaN-filled,
        // the lists have been enlarged since, and the binning asked for now is complete
        ws.late_overflows += 1;
        if (ws.strict) raise_status(rc, "gi2d_bin_tiles", have_gil);
        {
            // a Python warning like the ctypes binding's, on EVERY occurrence (the text differs
            // from one to the next, so the once-per-location filter does not hide the later
            // ones; the backward runs on an autograd thread without the interpreter lock)
            int64_t info[10] = {0};
            gi2d_workspace_info(ws.handle, info);
            std::ostringstream msg;
            msg << "gaussianimage_amd: an earlier frame overflowed a tile list -- its overflowed "
                   "tiles were NaN-filled, so a loss or PSNR computed from that image is NaN "
                   "(occurrence " << ws.late_overflows << " on this workspace, lists at generation "
                << info[2] << ", capacity now " << info[0] << " per tile). The lists have been "
                   "enlarged; this call is complete. Set GI2D_SYNC=1 (or strict on the workspace "
                   "to raise instead) to size the lists synchronously in every call.";
            py::gil_scoped_acquire gil;
            if (PyErr_WarnEx(PyExc_RuntimeWarning, msg.str().c_str(), 2) < 0)
                throw py::error_already_set();
        }
        rc = GI2D_OK;
    }
    check(rc, "gi2d_bin_tiles", have_gil);
    ws.key = key;
    ws.keep_xys = xys;
    ws.keep_radii = radii;
}

// `t` as a detached contiguous float32 tensor (itself when it already is one).
inline at::Tensor f32c(const at::Tensor &t)
{
    at::Tensor d = t.requires_grad() ? t.detach() : t;
    if (d.scalar_type() == at::kFloat && d.is_contiguous()) return d;
    return d.to(at::kFloat).contiguous();
}

inline const float *fptr(const at::Tensor &t) { return t.numel() ? t.data_ptr<float>() : nullptr; }
inline float *fptr_mut(at::Tensor &t) { return t.numel() ? t.data_ptr<float>() : nullptr; }
inline int *iptr_mut(at::Tensor &t) { return t.numel() ? t.data_ptr<int>() : nullptr; }

void need_device(const char *name, std::initializer_list<const at::Tensor *> tensors)
{
    for (const at::Tensor *t : tensors)
        if (!t->is_cuda()) {
            std::ostringstream msg;
            msg << name << ": this operator runs on an AMD GPU through the HIP library only; got a "
                << t->device() << " tensor (there is no CPU fallback)";
            throw std::runtime_error(msg.str());
        }
}

void tile_bounds_check(const char *name, int H, int W, const py::sequence &tile_bounds, int &tx,
                       int &ty)
{
    tx = (W + GI2D_BLOCK - 1) / GI2D_BLOCK;
    ty = (H + GI2D_BLOCK - 1) / GI2D_BLOCK;
    if (py::len(tile_bounds) < 2 || py::int_(tile_bounds[0]).cast<long>() != tx ||
        py::int_(tile_bounds[1]).cast<long>() != ty) {
        std::ostringstream msg;
        msg << name << ": tile_bounds " << std::string(py::str(py::tuple(tile_bounds)))
            << " does not match a " << W << "x" << H << " image with " << GI2D_BLOCK << "x"
            << GI2D_BLOCK << " tiles (expected (" << tx << ", " << ty << ", 1))";
        throw py::value_error(msg.str());
    }
}

struct ProjOutputs {
    at::Tensor xys, depths, radii, conics, nth;
};

ProjOutputs alloc_proj(int64_t N, const at::Tensor &like)
{
    auto f = like.options().dtype(at::kFloat);
    auto i = like.options().dtype(at::kInt);
    return {at::empty({N, 2}, f), at::empty({N}, f), at::empty({N}, i), at::empty({N, 3}, f),
            at::empty({N}, i)};
}

// ---- projection, Cholesky covariance (gaussianimage_cholesky.py:77) ---------------------------
ProjOutputs project_cholesky_forward(const at::Tensor &m, const at::Tensor &L, int H, int W, int tx,
                                     int ty)
{
    const int64_t N = m.size(0);
    ProjOutputs o = alloc_proj(N, m);
    void *st = raw_stream(m);
    auto ws = workspace_of(m, st);
    ws->key.valid = false;
    check(gi2d_project_cholesky_fwd(fptr(m), fptr(L), (int)N, H, W, tx, ty, g_conv, fptr_mut(o.xys),
                                    fptr_mut(o.depths), iptr_mut(o.radii), fptr_mut(o.conics),
                                    iptr_mut(o.nth), ws->handle, st),
          "gi2d_project_cholesky_fwd", true);
    ws->set_counted(o.xys, o.radii, H, W);
    return o;
}

struct ProjectCholesky : public torch::autograd::Function<ProjectCholesky> {
    static variable_list forward(AutogradContext *ctx, const at::Tensor &means2d,
                                 const at::Tensor &L_elements, int64_t H, int64_t W, int64_t tx,
                                 int64_t ty)
    {
        at::Tensor m = f32c(means2d), L = f32c(L_elements);
        ProjOutputs o = project_cholesky_forward(m, L, (int)H, (int)W, (int)tx, (int)ty);
        ctx->save_for_backward({m, L});
        ctx->saved_data["H"] = H;
        ctx->saved_data["W"] = W;
        ctx->mark_non_differentiable({o.depths, o.radii, o.nth});
        return {o.xys, o.depths, o.radii, o.conics, o.nth};
    }

    static variable_list backward(AutogradContext *ctx, variable_list grads)
    {
        auto saved = ctx->get_saved_variables();
        const at::Tensor &m = saved[0], &L = saved[1];
        const int H = (int)ctx->saved_data["H"].toInt(), W = (int)ctx->saved_data["W"].toInt();
        at::Tensor v_xys = grads[0].defined() ? f32c(grads[0]) : at::zeros_like(m);
        at::Tensor v_conics = grads[3].defined() ? f32c(grads[3]) : at::zeros_like(L);
        at::Tensor v_m = at::empty_like(m), v_L = at::empty_like(L);
        check(gi2d_project_cholesky_bwd(fptr(m), fptr(L), (int)m.size(0), H, W, fptr(v_xys),
                                        fptr(v_conics), fptr_mut(v_m), fptr_mut(v_L), raw_stream(m)),
              "gi2d_project_cholesky_bwd", false);
        return {v_m, v_L, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

// ---- projection, rotation-scale covariance (gaussianimage_rs.py:75) ---------------------------
ProjOutputs project_rs_forward(const at::Tensor &m, const at::Tensor &s, const at::Tensor &r, int H,
                               int W, int tx, int ty)
{
    const int64_t N = m.size(0);
    ProjOutputs o = alloc_proj(N, m);
    void *st = raw_stream(m);
    auto ws = workspace_of(m, st);
    ws->key.valid = false;
    check(gi2d_project_rs_fwd(fptr(m), fptr(s), fptr(r), (int)N, H, W, tx, ty, g_conv,
                              fptr_mut(o.xys), fptr_mut(o.depths), iptr_mut(o.radii),
                              fptr_mut(o.conics), iptr_mut(o.nth), ws->handle, st),
          "gi2d_project_rs_fwd", true);
    ws->set_counted(o.xys, o.radii, H, W);
    return o;
}

struct ProjectRS : public torch::autograd::Function<ProjectRS> {
    static variable_list forward(AutogradContext *ctx, const at::Tensor &means2d,
                                 const at::Tensor &scales2d, const at::Tensor &rotation, int64_t H,
                                 int64_t W, int64_t tx, int64_t ty)
    {
        at::Tensor m = f32c(means2d), s = f32c(scales2d), r = f32c(rotation);
        ProjOutputs o = project_rs_forward(m, s, r, (int)H, (int)W, (int)tx, (int)ty);
        ctx->save_for_backward({m, s, r});
        ctx->saved_data["H"] = H;
        ctx->saved_data["W"] = W;
        ctx->mark_non_differentiable({o.depths, o.radii, o.nth});
        return {o.xys, o.depths, o.radii, o.conics, o.nth};
    }

    static variable_list backward(AutogradContext *ctx, variable_list grads)
    {
        auto saved = ctx->get_saved_variables();
        const at::Tensor &m = saved[0], &s = saved[1], &r = saved[2];
        const int H = (int)ctx->saved_data["H"].toInt(), W = (int)ctx->saved_data["W"].toInt();
        const int64_t N = m.size(0);
        at::Tensor v_xys = grads[0].defined() ? f32c(grads[0]) : at::zeros_like(m);
        at::Tensor v_conics = grads[3].defined() ? f32c(grads[3]) : at::zeros({N, 3}, m.options());
        at::Tensor v_m = at::empty_like(m), v_s = at::empty_like(s), v_r = at::empty_like(r);
        check(gi2d_project_rs_bwd(fptr(m), fptr(s), fptr(r), (int)N, H, W, fptr(v_xys),
                                  fptr(v_conics), fptr_mut(v_m), fptr_mut(v_s), fptr_mut(v_r),
                                  raw_stream(m)),
              "gi2d_project_rs_bwd", false);
        return {v_m, v_s, v_r, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

// ---- rasterize_gaussians_sum (gaussianimage_cholesky.py:81-82) --------------------------------
struct RasterInputs {
    at::Tensor xys, radii, conics, colors, opacity, background;
};

// The tile lists are keyed on the caller's own xys/radii tensors, so that the passes of one model
// forward (gaussianimage_cholesky.py:81,89,95) share one binning.
RasterInputs raster_inputs(const at::Tensor &xys, const at::Tensor &radii, const at::Tensor &conics,
                           const at::Tensor &colors, const at::Tensor &opacity,
                           const at::Tensor &background)
{
    RasterInputs in;
    in.xys = xys.requires_grad() ? xys.detach() : xys;
    if (in.xys.scalar_type() != at::kFloat || !in.xys.is_contiguous()) in.xys = f32c(xys);
    in.radii = (radii.scalar_type() == at::kInt && radii.is_contiguous())
                   ? radii
                   : radii.to(at::kInt).contiguous();
    in.conics = f32c(conics);
    in.colors = f32c(colors);
    in.opacity = f32c(opacity).reshape({-1});
    in.background = f32c(background);
    return in;
}

at::Tensor raster_forward(const RasterInputs &in, int H, int W)
{
    void *st = raw_stream(in.xys);
    auto ws = workspace_of(in.xys, st);
    bin(*ws, in.xys, in.radii, H, W, 0, false, st, true);
    at::Tensor out = at::empty({H, W, 3}, in.xys.options());
    check(gi2d_rasterize_sum_fwd(ws->handle, fptr(in.xys), fptr(in.conics), fptr(in.colors),
                                 fptr(in.opacity), fptr(in.background), (int)in.xys.size(0), H, W,
                                 g_conv, fptr_mut(out), st),
          "gi2d_rasterize_sum_fwd", true);
    return out;
}

struct RasterizeSum : public torch::autograd::Function<RasterizeSum> {
    static at::Tensor forward(AutogradContext *ctx, const at::Tensor &xys, const at::Tensor &radii,
                              const at::Tensor &conics, const at::Tensor &colors,
                              const at::Tensor &opacity, const at::Tensor &background, int64_t H,
                              int64_t W)
    {
        RasterInputs in = raster_inputs(xys, radii, conics, colors, opacity, background);
        at::Tensor out = raster_forward(in, (int)H, (int)W);
        ctx->save_for_backward({in.xys, in.radii, in.conics, in.colors, in.opacity});
        ctx->saved_data["H"] = H;
        ctx->saved_data["W"] = W;
        ctx->saved_data["opacity_shape"] = opacity.sizes().vec();
        return out;
    }

    static variable_list backward(AutogradContext *ctx, variable_list grads)
    {
        auto saved = ctx->get_saved_variables();
        const at::Tensor &xys = saved[0], &radii = saved[1], &con = saved[2], &col = saved[3],
                         &opa = saved[4];
        const int H = (int)ctx->saved_data["H"].toInt(), W = (int)ctx->saved_data["W"].toInt();
        const int64_t N = xys.size(0);
        at::Tensor v_out = f32c(grads[0]);
        void *st = raw_stream(xys);
        auto ws = workspace_of(xys, st);
        // reuses the forward's lists unless another frame replaced them
        bin(*ws, xys, radii, H, W, 0, false, st, false);
        auto f = xys.options();
        at::Tensor v_xy = at::empty({N, 2}, f), v_con = at::empty({N, 3}, f),
                   v_col = at::empty({N, 3}, f), v_opa = at::empty({N}, f);
        check(gi2d_rasterize_sum_bwd(ws->handle, fptr(xys), radii.numel() ? radii.data_ptr<int>() : nullptr,
                                     fptr(con), fptr(col), fptr(opa), (int)N, H, W, g_conv,
                                     fptr(v_out), fptr_mut(v_xy), fptr_mut(v_con), fptr_mut(v_col),
                                     fptr_mut(v_opa), st),
              "gi2d_rasterize_sum_bwd", false);
        return {v_xy, at::Tensor(), v_con, v_col,
                v_opa.reshape(ctx->saved_data["opacity_shape"].toIntVector()), at::Tensor(),
                at::Tensor(), at::Tensor()};
    }
};

inline bool records_grad(std::initializer_list<const at::Tensor *> tensors)
{
    if (!at::GradMode::is_enabled()) return false;
    for (const at::Tensor *t : tensors)
        if (t->requires_grad()) return true;
    return false;
}

// ---- the Python-facing functions: the reference's signatures -----------------------------------
py::tuple py_project_gaussians_2d(const at::Tensor &means2d, const at::Tensor &L_elements,
                                  int64_t H, int64_t W, const py::sequence &tile_bounds)
{
    if (means2d.dim() != 2 || means2d.size(1) != 2)
        throw py::value_error("means2d must have shape (N, 2)");
    if (L_elements.dim() != 2 || L_elements.size(0) != means2d.size(0) || L_elements.size(1) != 3)
        throw py::value_error("L_elements must have shape (N, 3)");
    need_device("project_gaussians_2d", {&means2d, &L_elements});
    int tx, ty;
    tile_bounds_check("project_gaussians_2d", (int)H, (int)W, tile_bounds, tx, ty);
    if (records_grad({&means2d, &L_elements})) {
        variable_list o = ProjectCholesky::apply(means2d, L_elements, H, W, (int64_t)tx, (int64_t)ty);
        return py::make_tuple(o[0], o[1], o[2], o[3], o[4]);
    }
    // under torch.no_grad() (the reference's FPS and test loops, train.py:242,258) nothing is recorded
    ProjOutputs o = project_cholesky_forward(f32c(means2d), f32c(L_elements), (int)H, (int)W, tx, ty);
    return py::make_tuple(o.xys, o.depths, o.radii, o.conics, o.nth);
}

py::tuple py_project_gaussians_2d_scale_rot(const at::Tensor &means2d, const at::Tensor &scales2d,
                                            const at::Tensor &rotation, int64_t H, int64_t W,
                                            const py::sequence &tile_bounds)
{
    const int64_t N = means2d.dim() ? means2d.size(0) : 0;
    if (means2d.dim() != 2 || means2d.size(1) != 2)
        throw py::value_error("means2d must have shape (N, 2)");
    if (scales2d.dim() != 2 || scales2d.size(0) != N || scales2d.size(1) != 2)
        throw py::value_error("scales2d must have shape (N, 2)");
    if (rotation.numel() != N) throw py::value_error("rotation must have shape (N, 1)");
    need_device("project_gaussians_2d_scale_rot", {&means2d, &scales2d, &rotation});
    int tx, ty;
    tile_bounds_check("project_gaussians_2d_scale_rot", (int)H, (int)W, tile_bounds, tx, ty);
    if (records_grad({&means2d, &scales2d, &rotation})) {
        variable_list o =
            ProjectRS::apply(means2d, scales2d, rotation, H, W, (int64_t)tx, (int64_t)ty);
        return py::make_tuple(o[0], o[1], o[2], o[3], o[4]);
    }
    ProjOutputs o = project_rs_forward(f32c(means2d), f32c(scales2d), f32c(rotation), (int)H,
                                       (int)W, tx, ty);
    return py::make_tuple(o.xys, o.depths, o.radii, o.conics, o.nth);
}

// (depths and num_tiles_hit belong to the reference's signature; the sum needs neither)
at::Tensor py_rasterize_sum(const at::Tensor &xys, const at::Tensor &radii, const at::Tensor &conics,
                            const at::Tensor &colors, const at::Tensor &opacity, int64_t H,
                            int64_t W, int64_t block_h, int64_t block_w,
                            const c10::optional<at::Tensor> &background_opt)
{
    if (colors.dim() != 2) throw py::value_error("colors must have dimensions (N, D)");
    if (opacity.dim() != 2) throw py::value_error("opacity must have dimensions (N, 1)");
    if (colors.size(1) != 3)
        throw py::value_error("this build rasterizes 3 colour channels; got " +
                              std::to_string(colors.size(1)));
    if (block_h != GI2D_BLOCK || block_w != GI2D_BLOCK)
        throw py::value_error("tile size is fixed at 16x16 (the reference's train.py:54)");
    need_device("rasterize_gaussians_sum", {&xys, &conics, &colors, &opacity});
    at::Tensor background =
        background_opt.has_value() ? *background_opt : at::ones({3}, xys.options().dtype(at::kFloat));
    if (background.size(0) != colors.size(1))
        throw py::value_error("background and colors must have the same number of channels");
    const int64_t N = xys.size(0);
    const std::pair<const char *, const at::Tensor *> rows[] = {
        {"radii", &radii}, {"conics", &conics}, {"colors", &colors}, {"opacity", &opacity}};
    for (const auto &r : rows)
        if (r.second->size(0) != N)
            throw py::value_error(std::string(r.first) + " has " +
                                  std::to_string(r.second->size(0)) + " rows, xys has " +
                                  std::to_string(N));
    if (records_grad({&xys, &conics, &colors, &opacity}))
        return RasterizeSum::apply(xys, radii, conics, colors, opacity, background, H, W);
    return raster_forward(raster_inputs(xys, radii, conics, colors, opacity, background), (int)H,
                          (int)W);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "compiled binding of the gi2d operators (see gaussianimage_amd/ops.py)";
    m.def("init", [](uintptr_t conv_address, py::object error_class) {
        g_conv = reinterpret_cast<const gi2d_conv *>(conv_address);
        g_error_class = error_class.release().ptr();  // (kept for the life of the process)
    });
    m.def("abi_version", []() { return gi2d_version(); });
    m.def("build_id", []() { return std::string(GI2D_TORCH_OPS_ID); });
    py::class_<WsState, std::shared_ptr<WsState>>(m, "WsState")
        .def_property_readonly("handle", &WsState::handle_value)
        .def_readonly("device", &WsState::device)
        .def_readwrite("late_overflows", &WsState::late_overflows)
        .def_readwrite("strict", &WsState::strict)
        .def_readwrite("sync_binning", &WsState::sync_binning)
        .def_property_readonly("has_lists", [](const WsState &s) { return s.key.valid; })
        .def("invalidate", &WsState::invalidate, py::arg("counted") = false)
        .def("set_counted", &WsState::set_counted)
        .def("bin", [](WsState &s, const at::Tensor &xys, const at::Tensor &radii, int H, int W,
                       int flags, bool force) {
            bin(s, xys, radii, H, W, flags, force, raw_stream(xys), true);
        });
    m.def("workspace", [](int device, uintptr_t stream) { return workspace(device, stream); });
    m.def("drop_workspace", [](int device, uintptr_t stream) {
        // the next call on that stream starts from a fresh workspace; the dropped one is
        // destroyed when its last user lets go of it
        std::lock_guard<std::mutex> hold(g_ws_lock);
        g_ws.erase({device, stream});
    });
    m.def("project_gaussians_2d", &py_project_gaussians_2d);
    m.def("project_gaussians_2d_scale_rot", &py_project_gaussians_2d_scale_rot);
    m.def("rasterize_gaussians_sum", &py_rasterize_sum);
}
