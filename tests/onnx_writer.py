"""A minimal ONNX (protobuf wire format) WRITER for tests: just enough of ModelProto / GraphProto / NodeProto /
TensorProto to lay a BERT state dict out the way exporters do, so cs_bert_params_from_onnx can be exercised at
any size without the `onnx` package.  Test infrastructure only — the product reads ONNX, it never writes it.
(The fixture tests/golden/bert_tiny_export.onnx comes from torch's own exporter, not from this file.)"""
import numpy as np

FLOAT, FLOAT16, BFLOAT16 = 1, 10, 16
UINT8, INT8 = 2, 3


def raw_tensor(name: str, arr: np.ndarray, dtype: int) -> bytes:
    """An integer initialiser (UINT8 / INT8) exactly as given."""
    arr = np.ascontiguousarray(arr, np.uint8 if dtype == UINT8 else np.int8)
    body = b"".join(_key(1, 0) + _varint(int(d)) for d in arr.shape)
    body += _key(2, 0) + _varint(dtype)
    body += _ld(8, name.encode())
    return body + _ld(9, arr.tobytes())


def quantize(w: np.ndarray, dtype: int, axis=None):
    """onnxruntime-style affine quantisation of a float matrix -> (q, scale, zero_point): per tensor, or per slice along
    `axis`; INT8 symmetric (zero point 0), UINT8 asymmetric."""
    w = np.asarray(w, np.float32)
    red = None if axis is None else tuple(i for i in range(w.ndim) if i != axis)
    lo = np.minimum(w.min(axis=red, keepdims=axis is not None), 0.0)
    hi = np.maximum(w.max(axis=red, keepdims=axis is not None), 0.0)
    if dtype == INT8:
        scale = np.maximum(np.maximum(-lo, hi) / 127.0, 1e-12).astype(np.float32)
        zp = np.zeros_like(scale, dtype=np.int8)
        q = np.clip(np.rint(w / scale), -127, 127).astype(np.int8)
    else:
        scale = np.maximum((hi - lo) / 255.0, 1e-12).astype(np.float32)
        zp = np.clip(np.rint(-lo / scale), 0, 255).astype(np.uint8)
        q = np.clip(np.rint(w / scale) + zp.astype(np.float32), 0, 255).astype(np.uint8)
    return q, scale.reshape(-1) if axis is not None else scale.reshape(()), zp.reshape(-1) if axis is not None else zp.reshape(())


def _varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _key(num: int, wire: int) -> bytes:
    return _varint((num << 3) | wire)


def _ld(num: int, payload: bytes) -> bytes:
    return _key(num, 2) + _varint(len(payload)) + payload


def tensor(name: str, arr: np.ndarray, dtype: int = FLOAT, packed_float_data: bool = False) -> bytes:
    arr = np.ascontiguousarray(arr, np.float32)
    body = b"".join(_key(1, 0) + _varint(int(d)) for d in arr.shape)
    body += _key(2, 0) + _varint(dtype)
    if dtype == FLOAT:
        payload = arr.astype("<f4").tobytes()
    elif dtype == FLOAT16:
        payload = arr.astype("<f2").tobytes()
    else:  # bfloat16: the high half of the f32 (round to nearest even)
        u = arr.view(np.uint32).astype(np.uint64)
        payload = (((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype("<u2")).tobytes()
    body += _ld(8, name.encode())
    body += _ld(4, payload) if (packed_float_data and dtype == FLOAT) else _ld(9, payload)
    return body


def node(op: str, inputs, outputs, name: str = "", attrs=()) -> bytes:
    body = b"".join(_ld(1, i.encode()) for i in inputs) + b"".join(_ld(2, o.encode()) for o in outputs)
    if name:
        body += _ld(3, name.encode())
    body += _ld(4, op.encode())
    for an, iv in attrs:  # integer attributes only
        body += _ld(5, _ld(1, an.encode()) + _key(3, 0) + _varint(iv) + _key(20, 0) + _varint(2))
    return body


def model(nodes, initializers, producer: str = "tests/onnx_writer.py") -> bytes:
    graph = b"".join(_ld(1, n) for n in nodes) + _ld(2, b"main_graph") + b"".join(_ld(5, t) for t in initializers)
    return _key(1, 0) + _varint(8) + _ld(2, producer.encode()) + _ld(7, graph)


def bert_onnx(sd: dict, layers: int, style: str = "matmul", dtype: int = FLOAT, prefix: str = "", qdtype: int = INT8,
              per_channel: bool = False, quantize_tables: bool = False, dequantized: dict = None) -> bytes:
    """sd: HF BertModel state dict (numpy).  style:
       "matmul" — the torch.onnx / optimum shape: biases and other directly-consumed parameters keep their names,
                  Linear weights are transposed anonymous `onnx::MatMul_N` initialisers feeding MatMul -> Add(bias);
       "gemm"   — Gemm(x, W [out, in], bias, transB = 1) with the weight's own name;
       "fused"  — ORT-optimised: one com.microsoft Attention node per layer with a packed [H, 3H] weight and [3H] bias;
       "quantized" — onnxruntime's dynamic quantisation of the "matmul" shape: W_quantized (qdtype, per tensor or per
                  output channel) + W_scale + W_zero_point consumed by DynamicQuantizeLinear -> MatMulInteger -> Cast ->
                  Mul(Mul(x_scale, W_scale)) -> Add(bias); with quantize_tables the word-embedding Gather table too.
                  `dequantized` (a dict) receives name -> the f32 values a reader must reproduce.
       "optimized" / "optimized_quantized" — what onnxruntime's transformer optimiser leaves (model_optimized.onnx; the
                  quantised form is e.g. the BGE-small *Q entry): the bias Adds are swallowed by fused com.microsoft nodes —
                  SkipLayerNormalization(input, skip, gamma, beta, bias) behind attention.output / output.dense,
                  BiasGelu(x, bias) behind intermediate.dense, Attention (QAttention when quantised: weight int8 [H, 3H],
                  weight_scale, weight_zero_point) for Q / K / V, EmbedLayerNormalization for the three tables."""
    inits, nodes, counter = [], [], [1000]
    optimized = style in ("optimized", "optimized_quantized")
    quantized = style in ("quantized", "optimized_quantized")

    def keep(name):
        inits.append(tensor(prefix + name, sd[name], dtype))

    def linear(base, x, skip=None):
        w, b = sd[base + ".weight"], base + ".bias"
        keep(b)
        y = "/" + base + "/out"

        def finish(mm):
            """the node that consumes the bias: Add, or the fused node of an optimised file"""
            if not optimized:
                nodes.append(node("Add", [prefix + b, mm] if counter[0] % 3 else [mm, prefix + b], [y]))
            elif skip is None:   # intermediate.dense
                nodes.append(node("BiasGelu", [mm, prefix + b], [y]))
            else:                # attention.output.dense / output.dense: + residual, LayerNorm, in one node
                ln = base.replace(".dense", ".LayerNorm")
                ins = [mm, skip] if counter[0] % 2 else [skip, mm]
                nodes.append(node("SkipLayerNormalization", ins + [prefix + ln + ".weight", prefix + ln + ".bias", prefix + b], [y]))

        if style == "gemm":
            keep(base + ".weight")
            nodes.append(node("Gemm", [x, prefix + base + ".weight", prefix + b], [y], attrs=[("transB", 1)]))
        elif quantized:
            counter[0] += 1
            anon = f"onnx::MatMul_{counter[0]}"
            q, sc, zp = quantize(w.T, qdtype, axis=1 if per_channel else None)   # stored [in, out]; channels = outputs
            inits.append(raw_tensor(anon + "_quantized", q, qdtype))
            inits.append(tensor(anon + "_scale", sc))
            inits.append(raw_tensor(anon + "_zero_point", zp, qdtype))
            if dequantized is not None:
                dequantized[base + ".weight"] = ((q.astype(np.float32) - zp.astype(np.float32)) * sc).T.astype(np.float32)
            xq, xs, xz = (f"/{base}/x_{t}" for t in ("quantized", "scale", "zero_point"))
            nodes.append(node("DynamicQuantizeLinear", [x], [xq, xs, xz]))
            nodes.append(node("MatMulInteger", [xq, anon + "_quantized", xz, anon + "_zero_point"], [f"/{base}/mmi"]))
            nodes.append(node("Cast", [f"/{base}/mmi"], [f"/{base}/mmi_f"], attrs=[("to", 1)]))
            nodes.append(node("Mul", [xs, anon + "_scale"], [f"/{base}/scales"]))
            nodes.append(node("Mul", [f"/{base}/mmi_f", f"/{base}/scales"] if counter[0] % 2 else [f"/{base}/scales", f"/{base}/mmi_f"],
                              [f"/{base}/mm"]))
            finish(f"/{base}/mm")
        else:
            counter[0] += 1
            anon = f"onnx::MatMul_{counter[0]}"
            inits.append(tensor(anon, w.T, dtype, packed_float_data=(counter[0] % 2 == 0)))
            mm = "/" + base + "/MatMul_output_0"
            nodes.append(node("MatMul", [x, anon], [mm]))
            finish(mm)
        return y

    for n in ("embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight",
              "embeddings.token_type_embeddings.weight", "embeddings.LayerNorm.weight", "embeddings.LayerNorm.bias"):
        if quantized and quantize_tables and n == "embeddings.word_embeddings.weight":
            q, sc, zp = quantize(sd[n], UINT8)
            inits.append(raw_tensor(prefix + n + "_quantized", q, UINT8))
            inits.append(tensor(prefix + n + "_scale", sc))
            inits.append(raw_tensor(prefix + n + "_zero_point", zp, UINT8))
            if dequantized is not None:
                dequantized[n] = ((q.astype(np.float32) - zp.astype(np.float32)) * sc).astype(np.float32)
            continue
        keep(n)
    if optimized and not quantize_tables:
        e = "embeddings."
        nodes.append(node("EmbedLayerNormalization", ["input_ids", "token_type_ids", prefix + e + "word_embeddings.weight",
                                                      prefix + e + "position_embeddings.weight", prefix + e + "token_type_embeddings.weight",
                                                      prefix + e + "LayerNorm.weight", prefix + e + "LayerNorm.bias", "attention_mask"],
                          ["/emb", "mask_index"]))
    else:
        nodes.append(node("Gather", [prefix + "embeddings.word_embeddings.weight", "input_ids"], ["/emb"]))
    x = "/emb"
    for l in range(layers):
        p = f"encoder.layer.{l}."
        if style == "fused" or optimized:
            wq, wk, wv = (sd[p + f"attention.self.{n}.weight"] for n in ("query", "key", "value"))
            bq, bk, bv = (sd[p + f"attention.self.{n}.bias"] for n in ("query", "key", "value"))
            packed = np.concatenate([wq.T, wk.T, wv.T], axis=1)   # [H, 3H]
            inits.append(tensor(f"Attention_{l}_qkv_bias", np.concatenate([bq, bk, bv]), dtype))
            if quantized:
                q, sc, zp = quantize(packed, qdtype, axis=1 if per_channel else None)
                inits.append(raw_tensor(f"Attention_{l}_qkv_weight_quantized", q, qdtype))
                inits.append(tensor(f"Attention_{l}_qkv_weight_scale", sc))
                inits.append(raw_tensor(f"Attention_{l}_qkv_weight_zero_point", zp, qdtype))
                if dequantized is not None:
                    deq = ((q.astype(np.float32) - zp.astype(np.float32)) * sc).astype(np.float32)
                    H = wq.shape[0]
                    for i, nme in enumerate(("query", "key", "value")):
                        dequantized[p + f"attention.self.{nme}.weight"] = deq[:, i * H:(i + 1) * H].T.copy()
                xq, xs, xz = (f"/att{l}/x_{t}" for t in ("quantized", "scale", "zero_point"))
                nodes.append(node("DynamicQuantizeLinear", [x], [xq, xs, xz]))
                nodes.append(node("QAttention", [xq, f"Attention_{l}_qkv_weight_quantized", f"Attention_{l}_qkv_bias", xs,
                                                 f"Attention_{l}_qkv_weight_scale", "mask_index", xz,
                                                 f"Attention_{l}_qkv_weight_zero_point"], [f"/att{l}"], attrs=[("num_heads", 12)]))
            else:
                inits.append(tensor(f"Attention_{l}_qkv_weight", packed, dtype))
                nodes.append(node("Attention", [x, f"Attention_{l}_qkv_weight", f"Attention_{l}_qkv_bias", "mask_index"],
                                  [f"/att{l}"], attrs=[("num_heads", 12)]))
            ctx = f"/att{l}"
        else:
            q = linear(p + "attention.self.query", x)
            k = linear(p + "attention.self.key", x)
            v = linear(p + "attention.self.value", x)
            nodes.append(node("Softmax", [q, k, v], [f"/ctx{l}"]))
            ctx = f"/ctx{l}"
        x = linear(p + "attention.output.dense", ctx, skip=x)
        keep(p + "attention.output.LayerNorm.weight")
        keep(p + "attention.output.LayerNorm.bias")
        h = linear(p + "intermediate.dense", x)
        x = linear(p + "output.dense", h, skip=x)
        keep(p + "output.LayerNorm.weight")
        keep(p + "output.LayerNorm.bias")
    return model(nodes, inits)


def nomic_onnx(sd: dict, layers: int, quantized: bool = False, qdtype: int = UINT8, per_channel: bool = False,
               gate_first: bool = False, dtype: int = FLOAT, prefix: str = "", dequantized: dict = None) -> bytes:
    """sd: a NomicBert state dict under the model repository's names (emb_ln, encoder.layers.N.attn.Wqkv / out_proj, norm1,
    mlp.fc11 / fc12 / fc2, norm2; no Linear biases).  The layout torch's exporter gives such a module (the real thing:
    tests/golden/nomic_tiny_export.onnx): embedding tables and LayerNorm parameters under their names, every Linear weight an
    anonymous transposed `onnx::MatMul_N` on a MatMul's second input, the attention products as MatMuls between
    activations, silu as Sigmoid + Mul.  quantized: onnxruntime's dynamic quantisation of it (DynamicQuantizeLinear ->
    MatMulInteger -> Cast -> Mul(scales); `dequantized` receives name -> the f32 weights a reader must reproduce).
    gate_first: fc12's product is emitted before fc11's (a reader must find the gate by its Sigmoid, not by position)."""
    inits, nodes, counter = [], [], [2000]

    def keep(name):
        inits.append(tensor(prefix + name, sd[name], dtype))

    def product(base, x):
        w = sd[base + ".weight"]
        counter[0] += 1
        anon = f"onnx::MatMul_{counter[0]}"
        y = f"/{base}/MatMul_output_0"
        if not quantized:
            inits.append(tensor(anon, w.T, dtype, packed_float_data=(counter[0] % 2 == 0)))
            nodes.append(node("MatMul", [x, anon], [y]))
            return y
        q, sc, zp = quantize(w.T, qdtype, axis=1 if per_channel else None)   # stored [in, out]; channels = outputs
        inits.append(raw_tensor(anon + "_quantized", q, qdtype))
        inits.append(tensor(anon + "_scale", sc))
        inits.append(raw_tensor(anon + "_zero_point", zp, qdtype))
        if dequantized is not None:
            dequantized[base + ".weight"] = ((q.astype(np.float32) - zp.astype(np.float32)) * sc).T.astype(np.float32)
        xq, xs, xz = (f"/{base}/x_{t}" for t in ("quantized", "scale", "zero_point"))
        nodes.append(node("DynamicQuantizeLinear", [x], [xq, xs, xz]))
        nodes.append(node("MatMulInteger", [xq, anon + "_quantized", xz, anon + "_zero_point"], [f"/{base}/mmi"]))
        nodes.append(node("Cast", [f"/{base}/mmi"], [f"/{base}/mmi_f"], attrs=[("to", 1)]))
        nodes.append(node("Mul", [xs, anon + "_scale"], [f"/{base}/scales"]))
        nodes.append(node("Mul", [f"/{base}/mmi_f", f"/{base}/scales"] if counter[0] % 2 else [f"/{base}/scales", f"/{base}/mmi_f"], [y]))
        return y

    for n in ("embeddings.word_embeddings.weight", "embeddings.token_type_embeddings.weight", "emb_ln.weight", "emb_ln.bias"):
        keep(n)
    nodes.append(node("Gather", [prefix + "embeddings.word_embeddings.weight", "input_ids"], ["/we"]))
    nodes.append(node("Gather", [prefix + "embeddings.token_type_embeddings.weight", "token_type_ids"], ["/te"]))
    nodes.append(node("Add", ["/we", "/te"], ["/emb"]))
    nodes.append(node("LayerNormalization", ["/emb", prefix + "emb_ln.weight", prefix + "emb_ln.bias"], ["/x0"]))
    x = "/x0"
    for l in range(layers):
        p = f"encoder.layers.{l}."
        qkv = product(p + "attn.Wqkv", x)
        nodes.append(node("Split", [qkv], [f"/q{l}", f"/k{l}", f"/v{l}"]))
        nodes.append(node("MatMul", [f"/q{l}", f"/k{l}"], [f"/s{l}"]))          # products between activations: no initialiser
        nodes.append(node("Softmax", [f"/s{l}"], [f"/p{l}"]))
        nodes.append(node("MatMul", [f"/p{l}", f"/v{l}"], [f"/ctx{l}"]))
        ao = product(p + "attn.out_proj", f"/ctx{l}")
        nodes.append(node("Add", [ao, x], [f"/r1_{l}"]))
        keep(p + "norm1.weight")
        keep(p + "norm1.bias")
        nodes.append(node("LayerNormalization", [f"/r1_{l}", prefix + p + "norm1.weight", prefix + p + "norm1.bias"], [f"/x1_{l}"]))
        x = f"/x1_{l}"
        if gate_first:
            g = product(p + "mlp.fc12", x)
            v = product(p + "mlp.fc11", x)
        else:
            v = product(p + "mlp.fc11", x)
            g = product(p + "mlp.fc12", x)
        nodes.append(node("Sigmoid", [g], [f"/sig{l}"]))
        nodes.append(node("Mul", [g, f"/sig{l}"], [f"/silu{l}"]))
        nodes.append(node("Mul", [v, f"/silu{l}"], [f"/gated{l}"]))
        down = product(p + "mlp.fc2", f"/gated{l}")
        nodes.append(node("Add", [down, x], [f"/r2_{l}"]))
        keep(p + "norm2.weight")
        keep(p + "norm2.bias")
        nodes.append(node("LayerNormalization", [f"/r2_{l}", prefix + p + "norm2.weight", prefix + p + "norm2.bias"], [f"/x2_{l}"]))
        x = f"/x2_{l}"
    return model(nodes, inits)
