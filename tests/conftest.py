import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# ---- parity margins on record: tests call `margins.add(name, value, bound)`; every value is printed in the terminal
# summary (so it lands in the driver's pytest log even under -q) and appended to gpurun_out/parity_margins.log
_MARGINS = []


class _Margins:
    def add(self, name, value, bound, unit=""):
        _MARGINS.append((name, float(value), float(bound), unit))


@pytest.fixture(scope="session")
def margins():
    return _Margins()


def pytest_terminal_summary(terminalreporter):
    if not _MARGINS:
        return
    terminalreporter.section("parity margins (measured / bound)")
    lines = [f"{name}: {value:.3e} / {bound:.1e} {unit}".rstrip() for name, value, bound, unit in _MARGINS]
    for ln in lines:
        terminalreporter.write_line(ln)
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_margins.log"), "a") as fh:
            fh.write("\n".join(lines) + "\n")
    except OSError:
        pass
