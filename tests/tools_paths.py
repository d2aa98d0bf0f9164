from curvis_amd.paths import write_orbit, write_through  # noqa: F401
