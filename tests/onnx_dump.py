"""A minimal reader of the ONNX protobuf wire format (test helper: lists a file's initialisers and nodes so that a test or a
fixture script can say what an exporter wrote; the `onnx` package is not installed here).  Not a product path: the product's
reader is codesearch_amd/csrc/onnx_reader.cpp.

    python tests/onnx_dump.py file.onnx        # initialisers (name, dtype, dims) and nodes (op: inputs -> outputs)
"""
import sys


def _varint(b, i):
    v = s = 0
    while True:
        c = b[i]
        i += 1
        v |= (c & 0x7F) << s
        if c < 0x80:
            return v, i
        s += 7


def fields(b):
    """(field number, wire type, value) of one message; length-delimited values are memoryviews"""
    i, n = 0, len(b)
    while i < n:
        k, i = _varint(b, i)
        num, wire = k >> 3, k & 7
        if wire == 0:
            v, i = _varint(b, i)
        elif wire == 1:
            v, i = b[i:i + 8], i + 8
        elif wire == 5:
            v, i = b[i:i + 4], i + 4
        elif wire == 2:
            ln, i = _varint(b, i)
            v, i = b[i:i + ln], i + ln
        else:
            raise ValueError(f"wire type {wire}")
        yield num, wire, v


def read(path):
    """{'initializers': [(name, dtype, dims)], 'nodes': [(op, [inputs], [outputs], name)]}"""
    data = memoryview(open(path, "rb").read())
    graph = next(v for num, _, v in fields(data) if num == 7)
    inits, nodes = [], []
    for num, _, v in fields(graph):
        if num == 5:  # TensorProto
            dims, dtype, name = [], 0, ""
            for fn, fw, fv in fields(v):
                if fn == 1:
                    if fw == 0:
                        dims.append(fv)
                    else:  # packed
                        j = 0
                        while j < len(fv):
                            d, j = _varint(fv, j)
                            dims.append(d)
                elif fn == 2:
                    dtype = fv
                elif fn == 8:
                    name = bytes(fv).decode()
            inits.append((name, dtype, dims))
        elif num == 1:  # NodeProto
            ins, outs, op, name = [], [], "", ""
            for fn, _, fv in fields(v):
                if fn == 1:
                    ins.append(bytes(fv).decode())
                elif fn == 2:
                    outs.append(bytes(fv).decode())
                elif fn == 3:
                    name = bytes(fv).decode()
                elif fn == 4:
                    op = bytes(fv).decode()
            nodes.append((op, ins, outs, name))
    return {"initializers": inits, "nodes": nodes}


if __name__ == "__main__":
    m = read(sys.argv[1])
    for name, dtype, dims in m["initializers"]:
        print("init", name, dtype, dims)
    for op, ins, outs, name in m["nodes"]:
        print("node", op, ins, "->", outs)
