"""ONE definition of the algorithmic bytes of the convolution-class launches (round-4 review: bench.py, tools/summarize_profiles.py and
tools/layer_table.py each had their own and quoted 4.087 / 4.002 / 3.95 GB for the same step).

Per layer, as SURVEY 8(d) counts them: the input and every fused addend read once, the weights read once per launch, the output written once,
at the storage size of the path (4 bytes fp32, 2 bytes bf16).  The figure is PER LAYER whatever the launch structure: a launch that keeps
several layers' intermediates on chip (the bf16 BasicBlock chains, csrc/conv_bf16_chain.hip) moves FEWER bytes than this, and its
counter / algorithmic ratio falls below 1 -- that is the point of fusing, not an accounting error.

`c` is one entry of GRNet.describe_convs(): cin == 0 marks the grouped launch of an HR module's 1x1 fuse terms (fp32 path), whose add_elems
is the number of floats it reads per frame and which writes outputs 0 .. nb-2 (nb = c["n_add"])."""


def conv_algorithmic_bytes(c, n, elem_bytes):
    """(bytes read, bytes written) of one convolution-class layer in a call of n frames."""
    if c["cin"]:
        read = elem_bytes * (n * (c["cin"] * c["hin"] * c["win"] + c["add_elems"]) + c["ks"] * c["ks"] * c["cin"] * c["cout"])
        written = elem_bytes * n * c["cout"] * c["hout"] * c["wout"]
    else:
        nb = c["n_add"]
        read = elem_bytes * n * c["add_elems"]
        written = elem_bytes * n * sum((32 << i) * (56 >> i) ** 2 for i in range(nb - 1))
    return float(read), float(written)


def step_algorithmic_bytes(convs, n, elem_bytes):
    """Sum over describe_convs() of read + written bytes: the denominator of every traffic / algorithmic ratio in bench.py and profiles/."""
    return sum(sum(conv_algorithmic_bytes(c, n, elem_bytes)) for c in convs)
