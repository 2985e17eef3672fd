// ORACLE — TEST INFRASTRUCTURE ONLY.  placeholder, filled in below.
