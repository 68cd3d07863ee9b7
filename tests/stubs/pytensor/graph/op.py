class Op:
    """Only what ``perform`` needs: nothing."""
