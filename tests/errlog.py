"""Achieved error beside its bound: printed (pytest -s / failure reports) and appended to
gpurun_out/achieved_errors.txt, so that a drift inside a bound shows up in the run's record."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LOG = os.path.join(ROOT, 'gpurun_out', 'achieved_errors.txt')


def held(name, err, bound):
    """Asserts err <= bound and records both."""
    line = "%-72s achieved %.3e   bound %.1e" % (name, float(err), float(bound))
    print(line)
    try:
        os.makedirs(os.path.dirname(_LOG), exist_ok=True)
        with open(_LOG, 'a') as f:
            f.write(line + "\n")
    except OSError:
        pass
    assert err <= bound, line
