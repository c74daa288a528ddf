"""Host-side native runtime (C++, no device code): shared-memory RPC ring."""
