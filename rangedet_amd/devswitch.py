"""Development (A/B) switches of the lowering and the scheduler.

A release process takes none of them: what gets fused, folded or paired does not depend on the environment of whoever started the
process.  With RD_DEV_SWITCHES=1 in the environment the historical variables (RD_NO_FUSE_BLOCK, RD_PAIR, ... -- DESIGN.md
section 9) are honoured again, for A/B runs of one change on one box.  The native library follows the same rule at build time
(-DRD_DEV_SWITCHES, csrc/rd_common.h)."""
import os


def enabled():
    return os.environ.get("RD_DEV_SWITCHES", "0") not in ("", "0")


def get(name, default=None):
    return os.environ.get(name, default) if enabled() else default
