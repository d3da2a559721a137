"""polus.ir -> polus_amd.ir (re-export)."""
