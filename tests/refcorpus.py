"""The deterministic 190 100-byte differential corpus of the reference's compat test.

Data restated from meta/stdlib_compat_test.go:146-199 (`generateTestInput`: 41 fixed log-like
lines, each followed by '\\n', repeated 100 times).  Used by the golden tests only.
"""

LINES = [
    "HTTP/1.1 200 OK",
    "GET /api/users HTTP/1.1",
    "POST /api/login HTTP/1.1",
    "DELETE /api/session HTTP/1.1",
    "PUT /api/config HTTP/1.1",
    "PATCH /api/profile HTTP/1.1",
    '192.168.1.100 - - [15/Jan/2024:10:30:45 +0300] "GET /index.html HTTP/1.1" 200 1234',
    '10.0.0.1 - admin [15/Jan/2024:14:22:33 +0300] "POST /login?password=secret HTTP/1.1" 302 0',
    '172.16.0.50 - - [15/Jan/2024:23:59:59 +0300] "GET /api/data HTTP/1.1" 200 5678',
    "error: connection refused to database",
    "warning: disk space low on /dev/sda1",
    "fatal: unable to allocate memory",
    "critical: security breach detected",
    "[error] exception in handler: panic at line 42",
    "User-Agent: Mozilla/5.0 (Googlebot/2.1; +http://www.google.com/bot.html)",
    "User-Agent: Mozilla/5.0 (compatible; Bingbot/2.0; +http://www.bing.com/bingbot.htm)",
    "User-Agent: Mozilla/5.0 (compatible; YandexBot/3.0; +http://yandex.com/bots)",
    "user@example.com sent email to admin@company.org",
    "contact support+help@test-domain.co.uk for help",
    "Visit https://example.com/path?query=1#section or http://test.org/page",
    "Version 1.2.3 released, upgrading from 10.0.1 to 10.0.2",
    "eval(base64_decode('malicious')) detected in /var/www/uploads/shell.php",
    "phpinfo() call from 192.168.1.50 blocked",
    "SELECT * FROM users WHERE id=1 UNION SELECT password FROM admin",
    "Attempt to access /etc/passwd and ../../config",
    "apple banana cherry date elderberry fig grape honeydew kiwi lemon mango orange",
    "session_id=abc123def456 auth_token=0123456789abcdef0123456789abcdef",
    "login attempt for user admin from 10.0.0.5 at 08:15:30",
    "File report.txt created, also backup.log and notes.md available",
    "/admin/dashboard.php loaded in 0.5s",
    "/api/v2/users.php returned 404",
    "GET /static/style.css HTTP/1.1",
    "HEAD /health HTTP/1.1",
    "OPTIONS /api/cors HTTP/1.1",
    "abc123 def456 test789 hello world123",
    "word word2 word34 word567 word8901",
    "test testing tested tester",
    "line one here",
    "line two here",
    "hello Hello HELLO hElLo",
    "long token: " + "abcdef0123456789" * 3 + " end",
]


def generate_test_input() -> bytes:
    return ("".join(line + "\n" for line in LINES) * 100).encode()


# Patterns of the reference's TestStdlibCompatibility table (meta/stdlib_compat_test.go:27-67)
# that lay inside the restated subset of rounds 1-3 (no '.', no negated / Unicode classes); COMPAT_PATTERNS_WIDE below has more.
COMPAT_PATTERNS = {
    "literal_alt": r"error|warning|fatal|critical",
    "multi_literal": r"apple|banana|cherry|date|elderberry|fig|grape|honeydew|kiwi|lemon|mango|orange",
    "char_class": r"[\w]+",
    "email": r"[\w.+-]+@[\w.-]+\.[\w.-]+",
    "version": r"\d+\.\d+\.\d+",
    "ip": r"(?:(?:25[0-5]|2[0-4][0-9]|1[0-9][0-9]|[1-9]?[0-9])\.){3}(?:25[0-5]|2[0-4][0-9]|1[0-9][0-9]|[1-9]?[0-9])",
    "alpha_digit": r"[a-zA-Z]+\d+",
    "word_digit": r"\w+[0-9]+",
    "http_methods": r"(?m)^(GET|POST|PUT|DELETE|PATCH)",
    "word_repeat": r"(\w{2,8})+",
    "la_ips": r"\d+\.\d+\.\d+\.\d+",
    "la_emails": r"[\w.+-]+@[\w-]+\.[\w.-]+",
    "la_tokens": r"[a-f0-9]{32,}",
    "la_peak_hours": r"(?:0[0-9]|1[0-9]|2[0-3]):[0-5][0-9]:[0-5][0-9]",
    "word_boundary": r"\btest\b",
    "non_greedy": r"a+?",
    "alternation_overlap": r"ab|abc",
    "nested_groups": r"((a+)(b+))",
    "multiline_anchor": r"(?m)^line",
    "error_literal": r"error",
    "email_captures": r"(\w+)@(\w+)\.(\w+)",
}


# More rows of the same table, expressible since `.` and negated classes are restated (late round 3).  CPU tier only: the oracle is
# pinned on them, and the device twins where the product serves the pattern; `a*` and `.*` (empty matches) stay out.
COMPAT_PATTERNS_WIDE = {
    "anchored": r"^HTTP/[12]\.[01]",
    "inner_literal": r".*@example\.com",
    "suffix": r".*\.(txt|log|md)",
    "uri": r"[\w]+://[^/\s?#]+[^\s?#]+(?:\?[^\s#]*)?(?:#[^\s]*)?",
    "anchored_php": r"^/.*[\w-]+\.php",
    "multiline_php": r"(?m)^/.*\.php",
    "la_api_calls": r"(?m)^(?:GET|POST|PUT|DELETE|PATCH)\s+/api/\S+",
    "la_post_requests": r"(?m)^POST\s+\S+",
    "la_methods": r"(?m)^(GET|POST|PUT|DELETE|PATCH|HEAD|OPTIONS)\s",
    "la_passwords": r"(?m)^(?:GET|POST)\s+\S*(?:password|passwd|pwd|pass)\S*",
    "la_sessions": r"(?m)^(?:GET|POST)\s+\S*session\S*",
    # ... and the (?i) rows, since the case-fold literal expansion is restated (literal/extractor.go:838-941)
    "la_errors": r"(?i)(error|fail|exception|panic|fatal)",
    "la_bots": r"(?i)(googlebot|bingbot|yandexbot|baiduspider|duckduckbot|slurp|facebookexternalhit|twitterbot|rogerbot|linkedinbot|embedly|quora link preview|showyoubot|outbrain|pinterest|applebot|semrushbot|ahrefsbot|mj12bot|dotbot|petalbot|bytespider)",
    "la_suspicious": r"(?i)(eval|system|exec|execute|passthru|shell_exec|phpinfo|base64_decode|edoced_46esab|rot13|str_rot13|chmod|mkdir|fopen|fclose|readfile|union\s+select|etc/passwd|wp-admin|\.\./)",
    "la_auth_attempts": r"(?i)(?:login|auth|sign.?in|session)",
    "case_insensitive": r"(?i)hello",
}


def span_hash(spans) -> int:
    """Order-sensitive 64-bit FNV-1a over the int64 little-endian span stream."""
    import numpy as np

    a = np.ascontiguousarray(np.asarray(spans, dtype=np.int64)).tobytes()
    h = 0xCBF29CE484222325
    for b in a:
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h
