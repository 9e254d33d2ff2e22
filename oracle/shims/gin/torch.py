"""TEST INFRASTRUCTURE ONLY -- empty oracle shim for gin.torch (rave/balancer.py:2)."""
