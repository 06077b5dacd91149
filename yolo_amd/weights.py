"""Darknet-order weight streams for YOLOv3 (host side, numpy only).

Layout (reference yolo_v3/utils/misc_utils.py:45-96, convert_weight.py:15-31): 5 x int32 header, then per
conv in call order of `yolov3.forward`: [beta, gamma, moving_mean, moving_variance] (BN convs) or
[bias] (detection convs), then the kernel as [Cout, Cin, kh, kw] float32.
"""
import numpy as np


def yolov3_conv_specs(class_num=80):
    """The 75 convs of yolov3.forward in creation order (layer_utils.py:23-65, model.py:41-64)."""
    specs = []

    def conv(cin, cout, k, s=1, bn=True, res_tail=False):
        specs.append(dict(cin=cin, cout=cout, k=k, stride=s, bn=bn, res_tail=res_tail))
        return cout

    def res(c, f):
        conv(c, f, 1)
        return conv(f, 2 * f, 3, res_tail=True)

    c = conv(3, 32, 3)
    c = conv(c, 64, 3, 2)
    c = res(c, 32)
    c = conv(c, 128, 3, 2)
    for _ in range(2):
        c = res(c, 64)
    c = conv(c, 256, 3, 2)
    for _ in range(8):
        c = res(c, 128)
    c = conv(c, 512, 3, 2)
    for _ in range(8):
        c = res(c, 256)
    c = conv(c, 1024, 3, 2)
    for _ in range(4):
        c = res(c, 512)
    nout = 3 * (5 + class_num)
    for cin, f in ((1024, 512), (768, 256), (384, 128)):
        for a, b, k in ((cin, f, 1), (f, 2 * f, 3), (2 * f, f, 1), (f, 2 * f, 3), (2 * f, f, 1), (f, 2 * f, 3)):
            conv(a, b, k)
        conv(2 * f, nout, 1, bn=False)
        if f > 128:
            conv(f, f // 2, 1)
    return specs


def stream_length(specs):
    return sum((4 if s["bn"] else 1) * s["cout"] + s["cout"] * s["cin"] * s["k"] ** 2 for s in specs)


def synth_darknet_stream(class_num=80, seed=1234, res_gamma=0.2, head_gain=0.55, gamma_scale=0.9):
    """Seeded random-init weights in Darknet order (there are no trained weights offline).

    He-normal kernels; BN gamma~U(0.5,1.5)*gamma_scale (x res_gamma on convs feeding a residual add so
    that 23 residual adds do not blow the activations up), beta, mean ~N(0,0.1), var~U(0.5,1.5); head
    kernels x head_gain, head bias N(0,0.1) with the three objectness biases at -4.
    """
    rng = np.random.default_rng(seed)
    parts = []
    for s in yolov3_conv_specs(class_num):
        co, ci, k = s["cout"], s["cin"], s["k"]
        if s["bn"]:
            beta = rng.normal(0, 0.1, co)
            gamma = rng.uniform(0.5, 1.5, co) * (res_gamma if s["res_tail"] else gamma_scale)
            mean = rng.normal(0, 0.1, co)
            var = rng.uniform(0.5, 1.5, co)
            parts += [beta, gamma, mean, var]
        else:
            bias = rng.normal(0, 0.1, co)
            bias[4::co // 3] = -4.0
            parts.append(bias)
        w = rng.normal(0, np.sqrt(2.0 / (k * k * ci)), (co, ci, k, k))
        if not s["bn"]:
            w *= head_gain
        parts.append(w.ravel())
    return np.concatenate(parts).astype(np.float32)


def split_stream(stream, class_num=80):
    """-> per conv dict(spec=..., w=HWIO kernel, beta/gamma/mean/var or bias), walking the stream like
    load_weights (misc_utils.py:59-95)."""
    out, ptr = [], 0
    for s in yolov3_conv_specs(class_num):
        co, ci, k = s["cout"], s["cin"], s["k"]
        d = dict(spec=s)
        names = ("beta", "gamma", "mean", "var") if s["bn"] else ("bias",)
        for name in names:
            d[name] = np.ascontiguousarray(stream[ptr:ptr + co])
            ptr += co
        n = co * ci * k * k
        d["w"] = np.ascontiguousarray(np.transpose(stream[ptr:ptr + n].reshape(co, ci, k, k), (2, 3, 1, 0)))
        ptr += n
        out.append(d)
    if ptr != len(stream):
        raise ValueError("weight stream has %d floats, architecture needs %d" % (len(stream), ptr))
    return out


def write_darknet_weights(path, stream, header=(0, 2, 0, 0, 0)):
    with open(path, "wb") as fp:
        np.asarray(header, np.int32).tofile(fp)
        np.asarray(stream, np.float32).tofile(fp)


# ---- checkpoint ("state") in the reference's TF variable naming ---------------------------------------------------
def tf_variable_prefixes(class_num=80):
    """Scope of every conv's variables in the graph the reference builds under `tf.variable_scope('yolov3')`
    (convert_weight.py:21-25): slim numbers the convs per scope in creation order - `darknet53_body/Conv`,
    `Conv_1` .. `Conv_51` (model.py:38-39), then `yolov3_head/Conv` .. `Conv_22` (model.py:40-64)."""
    specs = yolov3_conv_specs(class_num)
    out = []
    for i in range(len(specs)):
        scope, j = ("darknet53_body", i) if i < 52 else ("yolov3_head", i - 52)
        out.append("yolov3/%s/%s" % (scope, "Conv" if j == 0 else "Conv_%d" % j))
    return out


def stream_to_state(stream, class_num=80):
    """Darknet-order stream -> dict of arrays named and laid out like the reference's TF variables after
    load_weights (misc_utils.py:59-95): `<conv>/weights` HWIO, `<conv>/BatchNorm/{gamma,beta,moving_mean,
    moving_variance}` or `<conv>/biases`."""
    state = {}
    for prefix, d in zip(tf_variable_prefixes(class_num), split_stream(np.asarray(stream, np.float32), class_num)):
        state[prefix + "/weights"] = d["w"]
        if d["spec"]["bn"]:
            for key, name in (("gamma", "gamma"), ("beta", "beta"), ("mean", "moving_mean"), ("var", "moving_variance")):
                state[prefix + "/BatchNorm/" + name] = d[key]
        else:
            state[prefix + "/biases"] = d["bias"]
    return state


def state_to_stream(state, class_num=80):
    """Inverse of stream_to_state.  Names may carry a ':0' suffix and may lack the leading 'yolov3/'."""
    norm = {}
    for name, v in state.items():
        name = name[:-2] if name.endswith(":0") else name
        norm[name if name.startswith("yolov3/") else "yolov3/" + name] = np.asarray(v, np.float32)
    parts, used = [], set()

    def take(name, shape):
        if name not in norm:
            raise KeyError("state has no variable %r" % name)
        v = norm[name]
        if tuple(v.shape) != tuple(shape):
            raise ValueError("variable %r has shape %s, architecture needs %s" % (name, tuple(v.shape), tuple(shape)))
        used.add(name)
        return v

    for prefix, s in zip(tf_variable_prefixes(class_num), yolov3_conv_specs(class_num)):
        co = s["cout"]
        if s["bn"]:
            for name in ("beta", "gamma", "moving_mean", "moving_variance"):        # file order (misc_utils.py:68-69)
                parts.append(take(prefix + "/BatchNorm/" + name, (co,)).ravel())
        else:
            parts.append(take(prefix + "/biases", (co,)).ravel())
        w = take(prefix + "/weights", (s["k"], s["k"], s["cin"], co))
        parts.append(np.transpose(w, (3, 2, 0, 1)).ravel())                          # HWIO -> [Cout, Cin, kh, kw]
    extra = sorted(set(norm) - used)
    if extra:
        raise ValueError("state has %d variables the architecture does not: %s ..." % (len(extra), extra[:3]))
    return np.concatenate(parts).astype(np.float32)
