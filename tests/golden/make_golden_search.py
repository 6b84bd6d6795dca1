"""Search-stage fixture generators (g6-g10); filled in with the host-logic rows."""
GENERATORS = {}
