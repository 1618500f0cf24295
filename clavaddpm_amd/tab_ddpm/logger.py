"""Key-value logger with the three calls the ClavaDDPM pipeline makes on `tab_ddpm.logger`
(`pipeline_utils.py:649-653, 786-790, 817`): `logkv`, `logkv_mean`, `dumpkvs`.

The reference vendors OpenAI-baselines' 495-line logger (file/TensorBoard/MPI back ends, all unused by the pipeline);
this module keeps only the contract those call sites rely on - last-value keys, running-mean keys, and a dump that
prints one aligned table to stdout and clears the accumulators - so `from clavaddpm_amd.tab_ddpm import logger`
is a drop-in for them.  Observability only: nothing here touches the GPU.
"""
import sys

_last = {}        # key -> most recent value (logkv)
_mean = {}        # key -> [sum, count] (logkv_mean)
_out = sys.stdout


def configure(stream=None):
    """Send dumps to `stream` (default: stdout) and forget everything accumulated so far."""
    global _out
    _out = stream if stream is not None else sys.stdout
    _last.clear()
    _mean.clear()


def logkv(key, val):
    """Record `val` under `key`; a later call for the same key before the dump overwrites it."""
    _last[key] = val


def logkv_mean(key, val):
    """Average every value recorded under `key` until the next dump."""
    acc = _mean.setdefault(key, [0.0, 0])
    acc[0] += float(val)
    acc[1] += 1


def logkvs(d):
    for k, v in d.items():
        logkv(k, v)


def getkvs():
    """Current table: last values plus the running means."""
    table = dict(_last)
    for k, (s, n) in _mean.items():
        table[k] = s / n if n else float("nan")
    return table


def _fmt(v):
    if isinstance(v, float):
        return f"{v:.3g}"
    return str(v)


def dumpkvs():
    """Print the table (keys sorted, one `| key | value |` row each), clear it, and return what was printed as a dict."""
    table = getkvs()
    if table:
        rows = [(str(k), _fmt(v)) for k, v in sorted(table.items(), key=lambda kv: str(kv[0]))]
        wk = max(len(k) for k, _ in rows)
        wv = max(len(v) for _, v in rows)
        bar = "-" * (wk + wv + 7)
        lines = [bar] + [f"| {k.ljust(wk)} | {v.ljust(wv)} |" for k, v in rows] + [bar]
        _out.write("\n".join(lines) + "\n")
        _out.flush()
    _last.clear()
    _mean.clear()
    return table
