"""Names of the upfirdn2d fixture cases (kept in sync with oracle/make_golden.py UFD_CASES)."""
UFD_TAGS = ["blur_up", "blur_d22", "skip_up", "down_11", "down_22", "odd", "crop", "up3dn2", "big", "asym"]
