"""TEST INFRASTRUCTURE ONLY -- empty oracle shim."""
