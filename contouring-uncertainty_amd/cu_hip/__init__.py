"""Host side of the HIP extension: ctypes binding (lib), tensor-level wrappers (ops), kernel schedules (engine)."""
