import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_report_header(config):
    """The tolerance behind every `== exact` assertion of this run (0.0 = identical doubles) goes into the report header, so that a
    log shows whether the suite ran bit for bit or relaxed (tests/golden_util.py: exact_tolerance)."""
    lines = []
    try:
        import golden_util
        lines.append("tempestmodel_amd: " + golden_util.exact_tolerance_report())
    except Exception as ex:
        lines.append("tempestmodel_amd: exact-comparison tolerance UNAVAILABLE (%s)" % ex)
    try:      # the live reference (oracle/_ref, built from /root/reference in the build container): required on a GPU host (TMX_REQUIRE_LIVE_REF)
        import golden_util
        ref = os.path.join(ROOT, "oracle", "_ref", "tmx_ref_dump_lapack2")
        lines.append("tempestmodel_amd: live reference %s, %s" % ("present" if os.path.exists(ref) else "MISSING",
                     "required (a missing binary fails its tests)" if golden_util.live_reference_required() else "optional (its tests skip without it)"))
    except Exception:
        pass
    return lines


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The same line at the end of the run: -q suppresses the header, and the driver's logs are -q runs."""
    try:
        import golden_util
        if golden_util._EXACT_TOL is not None:      # only when a test of this run asked for it
            terminalreporter.write_line("tempestmodel_amd: " + golden_util.exact_tolerance_report())
        if golden_util.live_reference_required():      # (a GPU host: say so in -q logs too)
            ref = os.path.join(ROOT, "oracle", "_ref", "tmx_ref_dump_lapack2")
            terminalreporter.write_line("tempestmodel_amd: live reference %s and required (TMX_REQUIRE_LIVE_REF)" % ("present" if os.path.exists(ref) else "MISSING"))
    except Exception:
        pass
