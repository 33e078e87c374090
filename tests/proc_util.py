"""Child processes of the tests: a command that outlives its time limit is killed WITH its descendants (a launcher's
ranks, a bench's self-started torch.distributed.run) so that nothing keeps the GPU or a rendezvous port afterwards."""
import os
import signal
import subprocess


def run(cmd, timeout, **kw):
    """subprocess.run(cmd, capture_output=True, text=True, timeout=...) in its own session; on a time-out the whole
    process group gets SIGKILL before TimeoutExpired propagates."""
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True, **kw)
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        proc.communicate()
        raise
    return subprocess.CompletedProcess(cmd, proc.returncode, out, err)
