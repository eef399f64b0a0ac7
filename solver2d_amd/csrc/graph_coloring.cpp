// Greedy graph colouring and batch formation for the sweeps (host side).
#include "solver_internal.h"

double nowMs()
{
	using namespace std::chrono;
	return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

bool isPositionSolver(int type)
{
	return type == s2amd_solverPGS_NGS || type == s2amd_solverPGS_NGS_Block || type == s2amd_solverTGS_NGS || type == s2amd_solverXPBD;
}

// host copy of math.h:201-207 (same fp32 operations as the device helper)
bool rotIsFixedPoint(float s, float c)
{
	float mag = sqrtf(s * s + c * c);
	float invMag = mag > 0.0f ? 1.0f / mag : 0.0f;
	float ns = s * invMag, nc = c * invMag;
	return memcmp(&ns, &s, 4) == 0 && memcmp(&nc, &c, 4) == 0;
}

// Greedy colouring of a constraint graph.  edges[k] = {a, b} (b may equal -1 for one-body
// constraints); a body takes part in conflicts only when conflict[body] is true.  Constraints are
// visited in the given order and receive the lowest colour unused on both bodies, so the result is
// deterministic.  Returns colour per constraint and the colour count.
struct ColorMasks
{
	enum
	{
		WORDS = 4
	};
	std::vector<uint64_t> bits; // WORDS per body
	std::vector<std::vector<int>> overflow; // colours >= 64*WORDS (rare: bodies with hundreds of constraints)
};

// balanced > 0 (strip groups and resident islands: one constraint per thread and colour round, `balanced` threads): a repair pass after the greedy pass
// evens out colours wider than one workgroup.
int colorGraph(const std::vector<int>& ea, const std::vector<int>& eb, const std::vector<uint8_t>& conflict, int bodyCount,
			   std::vector<int>& color, int balanced, std::vector<uint64_t>* bitsOut)
{
	const int W = ColorMasks::WORDS;
	size_t n = ea.size();
	color.assign(n, 0);
	std::vector<int> population;
	std::vector<uint64_t> bits((size_t)bodyCount * W, 0);
	std::vector<std::vector<int>> extra;
	std::vector<int> extraIndex; // body -> index in extra or -1
	int colorCount = 0;
	for (size_t k = 0; k < n; ++k)
	{
		int a = ea[k], b = eb[k];
		bool ca = a >= 0 && conflict[a], cb = b >= 0 && b != a && conflict[b];
		int chosen = -1;
		for (int w = 0; w < W && chosen < 0; ++w)
		{
			uint64_t used = (ca ? bits[(size_t)a * W + w] : 0) | (cb ? bits[(size_t)b * W + w] : 0);
			if (~used)
			{
				chosen = w * 64 + __builtin_ctzll(~used);
			}
		}
		if (chosen < 0)
		{
			// all 256 fast colours taken on these bodies: linear probe in the overflow sets
			if (extraIndex.empty())
			{
				extraIndex.assign(bodyCount, -1);
			}
			auto usedIn = [&](int body, int c) {
				int ei = extraIndex[body];
				if (ei < 0)
				{
					return false;
				}
				const std::vector<int>& v = extra[ei];
				return std::find(v.begin(), v.end(), c) != v.end();
			};
			int c = 64 * W;
			while ((ca && usedIn(a, c)) || (cb && usedIn(b, c)))
			{
				c += 1;
			}
			chosen = c;
			auto mark = [&](int body) {
				if (extraIndex[body] < 0)
				{
					extraIndex[body] = (int)extra.size();
					extra.emplace_back();
				}
				extra[extraIndex[body]].push_back(chosen);
			};
			if (ca)
			{
				mark(a);
			}
			if (cb)
			{
				mark(b);
			}
		}
		else
		{
			if (ca)
			{
				bits[(size_t)a * W + chosen / 64] |= 1ull << (chosen % 64);
			}
			if (cb)
			{
				bits[(size_t)b * W + chosen / 64] |= 1ull << (chosen % 64);
			}
		}
		color[k] = chosen;
		colorCount = std::max(colorCount, chosen + 1);
		if (balanced)
		{
			if ((int)population.size() <= chosen)
			{
				population.resize((size_t)chosen + 1, 0);
			}
			population[(size_t)chosen] += 1;
		}
	}
	if (balanced && colorCount <= 64)
	{
		// repair pass: greedy fills the low colours first; move constraints out of colours wider than one
		// workgroup into the least populated colour that is free on both bodies (never adds a colour)
		const int cap = balanced; // the widest colour class a round can take (threads of the workgroup)
		for (size_t kk = n; kk-- > 0;)
		{
			int c = color[kk];
			if (population[(size_t)c] <= cap)
			{
				continue;
			}
			int a = ea[kk], b = eb[kk];
			bool ca = a >= 0 && conflict[a], cb = b >= 0 && b != a && conflict[b];
			uint64_t used = (ca ? bits[(size_t)a * W] : 0) | (cb ? bits[(size_t)b * W] : 0);
			int best = -1;
			for (int c2 = 0; c2 < colorCount; ++c2)
			{
				if (c2 != c && ((used >> c2) & 1ull) == 0 && population[(size_t)c2] < cap && (best < 0 || population[(size_t)c2] < population[(size_t)best]))
				{
					best = c2;
				}
			}
			if (best < 0)
			{
				continue;
			}
			if (ca)
			{
				bits[(size_t)a * W] = (bits[(size_t)a * W] & ~(1ull << c)) | (1ull << best);
			}
			if (cb)
			{
				bits[(size_t)b * W] = (bits[(size_t)b * W] & ~(1ull << c)) | (1ull << best);
			}
			color[kk] = best;
			population[(size_t)c] -= 1;
			population[(size_t)best] += 1;
		}
		// what the repair could not place: a colour class is an independent set and so is any part of it, so a class
		// that is still wider than one workgroup is cut into classes of at most `cap` (more rounds, same validity)
		const int before = colorCount;
		std::vector<int> kept((size_t)before, 0);
		std::vector<int> spill((size_t)before, -1); // the colour currently receiving class c's overflow
		std::vector<int> spillCount((size_t)before, 0);
		for (size_t k = 0; k < n; ++k)
		{
			int c = color[k];
			if (c >= before || population[(size_t)c] <= cap)
			{
				continue;
			}
			if (kept[(size_t)c] < cap)
			{
				kept[(size_t)c] += 1;
				continue;
			}
			if (spill[(size_t)c] < 0 || spillCount[(size_t)c] == cap)
			{
				spill[(size_t)c] = colorCount++;
				spillCount[(size_t)c] = 0;
			}
			color[k] = spill[(size_t)c];
			spillCount[(size_t)c] += 1;
		}
	}
	if (bitsOut)
	{
		bitsOut->swap(bits); // colours 0..255 in use per body (the unbalanced greedy pass only: nothing was moved afterwards)
	}
	return colorCount;
}

// stable counting sort of constraint ids by colour
void sortByColor(const std::vector<int>& ids, const std::vector<int>& color, int colorCount, std::vector<int>& order, std::vector<int>& offsets)
{
	offsets.assign((size_t)colorCount + 1, 0);
	for (size_t k = 0; k < ids.size(); ++k)
	{
		offsets[(size_t)color[k] + 1] += 1;
	}
	for (int c = 0; c < colorCount; ++c)
	{
		offsets[(size_t)c + 1] += offsets[c];
	}
	std::vector<int> cursor(offsets.begin(), offsets.end() - 1);
	order.resize(ids.size());
	for (size_t k = 0; k < ids.size(); ++k)
	{
		order[(size_t)cursor[color[k]]++] = ids[k];
	}
}

// Launch batches from colour offsets.  Colours are launched one kernel each; when the colouring
// has a long run of tiny high colours (a body with dozens of constraints forces one colour per
// constraint) that run becomes ONE sequential tail batch instead of dozens of launches.
bool makeBatches(const std::vector<int>& colorOffsets, std::vector<int>& batchOffsets, bool allowTail, int tinyColor)
{
	int n = (int)colorOffsets.size() - 1;
	batchOffsets.clear();
	if (n <= 0)
	{
		batchOffsets.push_back(0);
		return false;
	}
	int total = colorOffsets[n];
	// the tail starts at the first colour from which on EVERY colour is tiny (a launch would cost more
	// than sweeping its few constraints serially); it must replace at least kMinTailColors launches
	// (tinyColor: 32 where a colour is a LAUNCH; inside an LDS group a colour is a barrier -- about the time of three turns of the tail's
	// walk --, and a tail that began at the first colour of 32 constraints swept 130 of a card house's 161 constraints one after the other:
	// 1.35 ms per TGS_Soft step, r6)
	const int kTinyColor = tinyColor, kMinTailColors = 4;
	int tailColor = n;
	for (int c = n - 1; c >= 1; --c)
	{
		if (colorOffsets[(size_t)c + 1] - colorOffsets[c] > kTinyColor)
		{
			break;
		}
		tailColor = c;
	}
	if (n - tailColor < kMinTailColors || (!allowTail && n <= 8))
	{
		tailColor = n; // strip groups with a handful of colours run them as preloaded rounds, however small
	}
	for (int c = 0; c <= tailColor; ++c)
	{
		batchOffsets.push_back(colorOffsets[c]);
	}
	if (tailColor < n)
	{
		batchOffsets.push_back(total);
		return true;
	}
	return false;
}

uint64_t fnv(uint64_t h, const void* data, size_t n)
{
	const unsigned char* p = (const unsigned char*)data;
	for (size_t i = 0; i < n; ++i)
	{
		h ^= p[i];
		h *= 1099511628211ull;
	}
	return h;
}
