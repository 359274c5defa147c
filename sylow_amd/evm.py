"""Host-side mirror of the reference's EVM precompile adapter (examples/reth_bn128.rs:99-217):
`run_add`, `run_mul`, `run_pair` with the same argument meaning (input bytes, gas cost(s), gas limit) and
error behaviour (`PrecompileError` carrying the reference's `Error` variant names), batched: each takes a
LIST of inputs and returns a list of results (`bytes`) or `PrecompileError` instances, one GPU launch for
the whole list.  Padding, length and gas rules are applied here; decode/validation/arithmetic run on the GPU.
"""
from __future__ import annotations

import numpy as np

from .engine import Engine

ADD_INPUT_LEN = 128          # reth_bn128.rs:83
MUL_INPUT_LEN = 96           # reth_bn128.rs:87
PAIR_ELEMENT_LEN = 192       # reth_bn128.rs:92
BYZANTIUM_ADD_GAS_COST, BYZANTIUM_MUL_GAS_COST = 500, 40_000
BYZANTIUM_PAIR_PER_POINT, BYZANTIUM_PAIR_BASE = 80_000, 100_000


class PrecompileError(Exception):
    OUT_OF_GAS = "OutOfGas"
    NOT_A_MEMBER = "Bn128FieldPointNotAMember"
    FAILED_TO_CREATE = "Bn128AffineGFailedToCreate"
    PAIR_LENGTH = "Bn128PairLength"

    def __init__(self, kind):
        super().__init__(kind)
        self.kind = kind

    def __eq__(self, other):
        return isinstance(other, PrecompileError) and other.kind == self.kind

    def __hash__(self):
        return hash(self.kind)


_STATUS = {4: PrecompileError.NOT_A_MEMBER, 1: PrecompileError.FAILED_TO_CREATE, 2: PrecompileError.FAILED_TO_CREATE}


def _right_pad(b: bytes, n: int) -> bytes:
    return (b + bytes(n))[:n]       # right_pad::<N>: pad with zeros, truncate to N


def _fixed(engine: Engine, name: str, inputs, in_len: int, gas_cost: int, gas_limits):
    n = len(inputs)
    out = [None] * n
    live = [i for i in range(n) if gas_cost <= gas_limits[i]]
    for i in range(n):
        if gas_cost > gas_limits[i]:
            out[i] = PrecompileError(PrecompileError.OUT_OF_GAS)
    if live:
        blob = np.frombuffer(b"".join(_right_pad(inputs[i], in_len) for i in live), dtype=np.uint8)
        d_in = engine.to_device(blob)
        d_out, d_st = engine.empty((len(live) * 64,), np.uint8), engine.empty((len(live),), np.uint8)
        engine._call(name, d_in.ptr, d_out.ptr, d_st.ptr, len(live))
        res, st = d_out.download().tobytes(), d_st.download()
        for j, i in enumerate(live):
            out[i] = PrecompileError(_STATUS[int(st[j])]) if st[j] else res[64 * j:64 * j + 64]
    return out


def run_add(engine: Engine, inputs, gas_cost=BYZANTIUM_ADD_GAS_COST, gas_limits=None):
    """reth_bn128.rs:130-141"""
    gas_limits = gas_limits or [gas_cost] * len(inputs)
    return _fixed(engine, "sylow_hip_evm_ecadd_batch", inputs, ADD_INPUT_LEN, gas_cost, gas_limits)


def run_mul(engine: Engine, inputs, gas_cost=BYZANTIUM_MUL_GAS_COST, gas_limits=None):
    """reth_bn128.rs:143-158 (scalars >= r are reduced mod r as EIP-196 specifies; the reference unwraps and would panic)"""
    gas_limits = gas_limits or [gas_cost] * len(inputs)
    return _fixed(engine, "sylow_hip_evm_ecmul_batch", inputs, MUL_INPUT_LEN, gas_cost, gas_limits)


def run_pair(engine: Engine, inputs, pair_per_point_cost=BYZANTIUM_PAIR_PER_POINT, pair_base_cost=BYZANTIUM_PAIR_BASE, gas_limits=None):
    """reth_bn128.rs:160-217: returns 32-byte big-endian 0/1 per input, or a PrecompileError."""
    n = len(inputs)
    gas_used = [(len(b) // PAIR_ELEMENT_LEN) * pair_per_point_cost + pair_base_cost for b in inputs]
    gas_limits = gas_limits or gas_used
    out = [None] * n
    live = []
    for i, b in enumerate(inputs):
        if gas_used[i] > gas_limits[i]:
            out[i] = PrecompileError(PrecompileError.OUT_OF_GAS)
        elif len(b) % PAIR_ELEMENT_LEN != 0:
            out[i] = PrecompileError(PrecompileError.PAIR_LENGTH)
        else:
            live.append(i)
    if live:
        counts = [len(inputs[i]) // PAIR_ELEMENT_LEN for i in live]
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
        n_pairs = int(off[-1])
        blob = np.frombuffer(b"".join(inputs[i] for i in live) or b"\x00", dtype=np.uint8)
        d_in, d_off = engine.to_device(blob), engine.to_device(off)
        d_res, d_st = engine.empty((len(live),), np.uint8), engine.empty((len(live),), np.uint8)
        engine._call("sylow_hip_evm_ecpairing_batch", d_in.ptr, d_off.ptr, len(live), n_pairs, d_res.ptr, d_st.ptr)
        res, st = d_res.download(), d_st.download()
        for j, i in enumerate(live):
            out[i] = PrecompileError(_STATUS[int(st[j])]) if st[j] else int(res[j]).to_bytes(32, "big")
    return out
