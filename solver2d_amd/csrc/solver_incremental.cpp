// Incremental updates of the constraint-graph structure: a created contact gets a place in the existing sweep order
// instead of a rebuild (IncrementalGlobal in solver_internal.h says what is kept and why the launch sequence survives).
//
// The reference creates and destroys contacts every step (src/world.c:125-168 -> src/contact.c:137-229).  Destruction
// never touches the structure here (the entry lingers as a no-op until the next rebuild); this file handles creation:
// for a new contact between bodies a and b of the GLOBAL part
//   * colour = the lowest parallel colour batch that is unused on both writable bodies AND still has a free position;
//   * position = the lowest free position of that batch (its contactIndex goes from -1 to the contact's pool slot);
//   * its two entries (k << 1 | side) are inserted into the bodies' incidence lists at their place in sweep order, so the
//     body-centric warm start and the Jacobi apply keep adding in sweep order.
// The result is a proper colouring and a valid sweep order like any other: the oracle sweeps in it (parity link L2).
#include "solver_internal.h"

#include <cstddef>
#include <unordered_map>

namespace
{

struct Patcher
{
	s2amdSolver* s;
	IncrementalGlobal& inc;
	void word(const void* base, size_t index, uint32_t value)
	{
		unsigned long long addr = (unsigned long long)(uintptr_t)base + 4ull * index;
		inc.patches.push_back(make_uint4((unsigned int)(addr & 0xffffffffull), (unsigned int)(addr >> 32), value, 0u));
	}
	void range(int body) // adjRange[body] -> device
	{
		word(s->dAdjOffsets.p, 2 * (size_t)body, (uint32_t)inc.adjRange[(size_t)body].x);
		word(s->dAdjOffsets.p, 2 * (size_t)body + 1, (uint32_t)inc.adjRange[(size_t)body].y);
	}
	void listEntry(int e) { word(s->dAdjList.p, (size_t)e, (uint32_t)inc.adjList[(size_t)e]); }
	void heavyEntry(int i) { word(s->dAdjHeavy.p, (size_t)i, (uint32_t)inc.heavy[(size_t)i]); }
};

bool writable(const s2amdSolver* s, int body)
{
	return (s->hBodyFlags[(size_t)body] & (s->inc.solverClass == 1 ? S2F_WRITE_POS : S2F_WRITE_VEL)) != 0;
}

void heavyRemove(Patcher& p, int body)
{
	IncrementalGlobal& inc = p.inc;
	const int n = inc.heavy[0];
	for (int i = 1; i <= n; ++i)
	{
		if (inc.heavy[(size_t)i] == body)
		{
			inc.heavy[(size_t)i] = inc.heavy[(size_t)n];
			inc.heavy[0] = n - 1;
			p.heavyEntry(i);
			p.heavyEntry(0);
			return;
		}
	}
}

bool adjInsert(Patcher& p, int body, int key)
{
	IncrementalGlobal& inc = p.inc;
	int2& r = inc.adjRange[(size_t)body];
	if (r.y == inc.adjCapacity[(size_t)body])
	{
		// the list outgrew its slack: it moves to the end of the array with twice the room
		const int cap = std::max(8, 2 * inc.adjCapacity[(size_t)body]);
		if ((size_t)inc.adjUsed + (size_t)cap > inc.adjList.size())
		{
			inc.adjFailure = 1;
			return false;
		}
		for (int i = 0; i < r.y; ++i)
		{
			inc.adjList[(size_t)(inc.adjUsed + i)] = inc.adjList[(size_t)(r.x + i)];
			p.listEntry(inc.adjUsed + i);
		}
		r.x = inc.adjUsed;
		inc.adjCapacity[(size_t)body] = cap;
		inc.adjUsed += cap;
	}
	int at = r.y;
	while (at > 0 && inc.adjList[(size_t)(r.x + at - 1)] > key)
	{
		inc.adjList[(size_t)(r.x + at)] = inc.adjList[(size_t)(r.x + at - 1)];
		p.listEntry(r.x + at);
		at -= 1;
	}
	inc.adjList[(size_t)(r.x + at)] = key;
	p.listEntry(r.x + at);
	r.y += 1;
	p.range(body);
	if (r.y == S2_HEAVY_DEGREE + 1)
	{
		// from here on a whole wave walks this body's list (warmStartBodiesKernel, jacobiApplyKernel)
		const int n = inc.heavy[0];
		if (n + 1 >= (int)inc.heavy.size())
		{
			inc.adjFailure = 2;
			return false;
		}
		inc.heavy[(size_t)n + 1] = body;
		inc.heavy[0] = n + 1;
		p.heavyEntry(n + 1);
		p.heavyEntry(0);
	}
	return true;
}

void adjRemove(Patcher& p, int body, int key)
{
	IncrementalGlobal& inc = p.inc;
	int2& r = inc.adjRange[(size_t)body];
	int at = 0;
	while (at < r.y && inc.adjList[(size_t)(r.x + at)] != key)
	{
		at += 1;
	}
	if (at == r.y)
	{
		return;
	}
	for (int i = at; i + 1 < r.y; ++i)
	{
		inc.adjList[(size_t)(r.x + i)] = inc.adjList[(size_t)(r.x + i + 1)];
		p.listEntry(r.x + i);
	}
	r.y -= 1;
	p.range(body);
	if (r.y == S2_HEAVY_DEGREE)
	{
		heavyRemove(p, body); // back to the one-thread walk: both paths must never take the same body
	}
}

// ---- strips (solver_internal.h: IncrementalStrips) ----

// where a constraint between bodies a and b would live in the strips: table (0 interior, 1 seam), group, the bodies' local slots
static bool stripHomeWhy(const s2amdSolver* s, int a, int b, int& table, int& group, int& la, int& lb, const char*& why, int* mover = nullptr);

// Can `body` leave the strip that owns it?  Only a body nothing in the strips refers to: no constraint of any round on it, in no
// seam's body list (IncrementalStrips: a body that joins an island).
static bool stripBodyIsFree(const s2amdSolver* s, int body)
{
	const IncrementalStrips& m = s->stripInc;
	const int g = m.ownerStrip[(size_t)body];
	if (g < 0 || m.stripBodyBase.empty() || !writable(s, body))
	{
		return false;
	}
	if (m.roundMask[0][(size_t)(m.bodyOffset[0][(size_t)g] + m.ownerSlot[(size_t)body])] != 0u)
	{
		return false;
	}
	for (int sm = g - 1; sm <= g; ++sm)
	{
		if (sm >= 0 && sm < (int)m.seamGroupOf.size() && m.seamGroupOf[(size_t)sm] >= 0 && m.seamSlot[(size_t)m.seamGroupOf[(size_t)sm]].count(body) != 0)
		{
			return false;
		}
	}
	return true;
}

// ... and can strip `g` take one more body?  (The build's budgets hold S2_STRIP_ADOPT_SLACK more per strip; a list that has not moved
// yet needs room behind the table.)
static bool stripCanAdopt(const s2amdSolver* s, int g)
{
	const IncrementalStrips& m = s->stripInc;
	if (s->optStripAdopt == 0 || m.takeOnly || g < 0 || g >= (int)m.stripBodyCount.size() || m.adoptedBy[(size_t)g] >= S2_STRIP_ADOPT_SLACK || (int)s->hPersistDescs.size() != (int)m.stripBodyCount.size())
	{
		return false;
	}
	const int nb = m.stripBodyCount[(size_t)g];
	if (nb + 1 > S2_STRIP_BODY_CHUNKS * 256)
	{
		return false;
	}
	return nb < m.stripListCapacity[(size_t)g] || m.spareIdsNext + nb + S2_STRIP_ADOPT_SLACK <= m.spareIdsEnd;
}

bool stripHome(const s2amdSolver* s, int a, int b, int& table, int& group, int& la, int& lb, int* mover)
{
	const char* why = "";
	const bool ok = stripHomeWhy(s, a, b, table, group, la, lb, why, mover);
	static const bool debug = getenv("S2AMD_DEBUG_PLACE") != nullptr;
	if (!ok && debug)
	{
		const IncrementalStrips& m = s->stripInc;
		fprintf(stderr, "[s2amd] no strip home for (%d, %d): %s (strips %d, %d)\n", a, b, why, m.ownerStrip[(size_t)a], m.ownerStrip[(size_t)b]);
	}
	return ok;
}
static bool stripHomeWhy(const s2amdSolver* s, int a, int b, int& table, int& group, int& la, int& lb, const char*& why, int* mover)
{
	const IncrementalStrips& m = s->stripInc;
	const int sa = writable(s, a) ? m.ownerStrip[(size_t)a] : -1, sb = writable(s, b) ? m.ownerStrip[(size_t)b] : -1;
	if (mover)
	{
		*mover = -1;
	}
	if (sa < 0 && sb < 0)
	{
		why = "no strip owns either body";
		return false;
	}
	if (sa >= 0 && sb >= 0 && sa != sb)
	{
		// a body that joins the island: nothing refers to it yet, so it moves to the other body's strip and the constraint is an
		// interior one there (the caller performs the move: `mover`; la / lb are the slots AFTER it)
		auto joins = [&](int body, int to, int& slotOut) {
			if (!stripBodyIsFree(s, body) || !stripCanAdopt(s, to))
			{
				return false;
			}
			slotOut = m.stripBodyCount[(size_t)to];
			return true;
		};
		auto seamHome = [&]() {
			if (sa - sb != 1 && sb - sa != 1)
			{
				why = "strips not adjacent";
				return false;
			}
			const int sm = std::min(sa, sb);
			if (sm >= (int)m.seamGroupOf.size() || m.seamGroupOf[(size_t)sm] < 0)
			{
				why = "no seam group";
				return false;
			}
			table = 1, group = m.seamGroupOf[(size_t)sm];
			const auto& slots = m.seamSlot[(size_t)group];
			const auto ia = slots.find(a), ib = slots.find(b);
			if (ia != slots.end() && ib != slots.end())
			{
				la = ia->second, lb = ib->second;
				return true;
			}
			// a body the seam does not carry yet becomes its next local body (the caller appends it: `mover` = -2 - {1: a, 2: b, 3: both})
			const int missing = (ia == slots.end() ? 1 : 0) | (ib == slots.end() ? 2 : 0);
			int next = m.seamBodyCount.empty() ? 0 : m.seamBodyCount[(size_t)group];
			int extraLeft = 0, extraRight = 0, importsOf[2] = {0, 0}; // (what this placement adds: a and b sit on different sides)
			for (int which = 1; which <= 2 && mover; which <<= 1)
			{
				if ((missing & which) == 0)
				{
					continue;
				}
				const int owner = which == 1 ? sa : sb, other = which == 1 ? sb : sa, body = which == 1 ? a : b;
				const bool left = owner == sm;
				(left ? extraLeft : extraRight) += 1;
				importsOf[other == sm ? 0 : 1] += 1;
				// a body is carried by ONE seam at most: the two seams of a strip are swept in the same rounds, each by its own pair of
				// workgroups, and the neighbour on one side never sees what the other seam does to the body within a sweep
				const int otherSeam = left ? sm - 1 : sm + 1;
				if (otherSeam >= 0 && otherSeam < (int)m.seamGroupOf.size() && m.seamGroupOf[(size_t)otherSeam] >= 0 &&
					m.seamSlot[(size_t)m.seamGroupOf[(size_t)otherSeam]].count(body) != 0)
				{
					why = "the body is carried by its strip's other seam";
					return false;
				}
			}
			const bool room = mover && s->optStripAdopt != 0 && !m.takeOnly && (int)s->hPersistDescs.size() == (int)m.stripBodyCount.size() && !m.seamBodyCount.empty() &&
							  m.seamExtra[0][(size_t)group] + extraLeft <= S2_STRIP_ADOPT_SLACK && m.seamExtra[1][(size_t)group] + extraRight <= S2_STRIP_ADOPT_SLACK &&
							  m.adoptedBy[(size_t)sm] + importsOf[0] <= S2_STRIP_ADOPT_SLACK && m.adoptedBy[(size_t)sm + 1] + importsOf[1] <= S2_STRIP_ADOPT_SLACK &&
							  s->hPersistDescs[(size_t)sm].exportCount[1] + extraLeft <= 256 && s->hPersistDescs[(size_t)sm + 1].exportCount[0] + extraRight <= 256;
			if (!room)
			{
				why = "a body is not in the seam's body list";
				return false;
			}
			la = ia != slots.end() ? ia->second : next++;
			lb = ib != slots.end() ? ib->second : next++;
			*mover = -2 - missing;
			return true;
		};
		// the two sides of a seam: both bodies must already be in its body list (the exchange carries exactly those)
		if (seamHome())
		{
			return true;
		}
		if (mover && joins(b, sa, lb))
		{
			table = 0, group = sa, la = m.ownerSlot[(size_t)a], *mover = b;
			return true;
		}
		if (mover && joins(a, sb, la))
		{
			table = 0, group = sb, lb = m.ownerSlot[(size_t)b], *mover = a;
			return true;
		}
		return false;
	}
	// an interior constraint of the strip that owns the writable side(s); a read-only body must have its replica there already
	table = 0, group = sa >= 0 ? sa : sb;
	auto slotIn = [&](int body, int owner, int& out) {
		if (owner >= 0)
		{
			out = m.ownerSlot[(size_t)body];
			return true;
		}
		if (writable(s, body))
		{
			why = "a writable body no strip owns";
			return false; // (it would have to join the island)
		}
		const auto& rep = m.replicaSlot[(size_t)group];
		const auto it = rep.find(body);
		if (it == rep.end())
		{
			why = "read-only body without a replica in the strip";
			return false;
		}
		out = it->second;
		return true;
	};
	return slotIn(a, sa, la) && slotIn(b, sb, lb);
}

// `body` (stripBodyIsFree) becomes a body of strip `to` (stripCanAdopt): IncrementalStrips says what that takes
static void stripAdopt(s2amdSolver* s, Patcher& p, int body, int to)
{
	IncrementalStrips& m = s->stripInc;
	const int* ids = s->dStripA.view.bodyIds;
	const int from = m.ownerStrip[(size_t)body], fromSlot = m.ownerSlot[(size_t)body];
	// the old entry stays where it is, as a read-only copy nothing refers to
	std::vector<int>& fromMoved = m.movedList[(size_t)from];
	if (fromMoved.empty())
	{
		const HostGroupTable& h = s->hStripA;
		fromMoved.assign(h.bodyIds.begin() + h.bodyOffsets[(size_t)from], h.bodyIds.begin() + h.bodyOffsets[(size_t)from + 1]);
	}
	fromMoved[(size_t)fromSlot] = body;
	p.word(ids, (size_t)(m.stripBodyBase[(size_t)from] + fromSlot), (uint32_t)body);
	// the receiving list: behind the table on its first adoption (with room for the next ones)
	const int nb = m.stripBodyCount[(size_t)to];
	std::vector<int>& list = m.movedList[(size_t)to];
	if (nb >= m.stripListCapacity[(size_t)to])
	{
		const HostGroupTable& h = s->hStripA;
		if (list.empty())
		{
			list.assign(h.bodyIds.begin() + h.bodyOffsets[(size_t)to], h.bodyIds.begin() + h.bodyOffsets[(size_t)to + 1]);
		}
		const int base = m.spareIdsNext;
		m.spareIdsNext += nb + S2_STRIP_ADOPT_SLACK;
		m.stripBodyBase[(size_t)to] = base, m.stripListCapacity[(size_t)to] = nb + S2_STRIP_ADOPT_SLACK;
		for (int i = 0; i < nb; ++i)
		{
			p.word(ids, (size_t)(base + i), (uint32_t)list[(size_t)i]);
		}
		// roundMask: the strip's entries move behind the others the same way
		const int off = m.bodyOffset[0][(size_t)to], newOff = (int)m.roundMask[0].size();
		m.roundMask[0].resize((size_t)(newOff + nb + S2_STRIP_ADOPT_SLACK), 0u);
		std::copy(m.roundMask[0].begin() + off, m.roundMask[0].begin() + off + nb, m.roundMask[0].begin() + newOff);
		m.bodyOffset[0][(size_t)to] = newOff;
	}
	else if (list.empty())
	{
		const HostGroupTable& h = s->hStripA;
		list.assign(h.bodyIds.begin() + h.bodyOffsets[(size_t)to], h.bodyIds.begin() + h.bodyOffsets[(size_t)to + 1]);
	}
	list.resize((size_t)nb);
	list.push_back((int)((uint32_t)body | S2G_OWNED));
	p.word(ids, (size_t)(m.stripBodyBase[(size_t)to] + nb), (uint32_t)body | S2G_OWNED);
	m.roundMask[0][(size_t)(m.bodyOffset[0][(size_t)to] + nb)] = 0u;
	m.stripBodyCount[(size_t)to] = nb + 1;
	m.adoptedBy[(size_t)to] += 1;
	m.ownerStrip[(size_t)body] = to, m.ownerSlot[(size_t)body] = nb;
	// the descriptor: where the list is, how long it is
	static_assert(offsetof(StripDesc, bodyBase) == 0 && offsetof(StripDesc, bodyCount) == 4, "StripDesc layout");
	const size_t words = (size_t)((const uint32_t*)(s->leanA.descs + to) - (const uint32_t*)s->leanA.descs);
	p.word(s->leanA.descs, words + 0, (uint32_t)m.stripBodyBase[(size_t)to]);
	p.word(s->leanA.descs, words + 1, (uint32_t)(nb + 1));
	// the imports follow the own bodies in LDS: every seam-local body of this strip that is an import sits one slot further on
	const PersistDesc& d = s->hPersistDescs[(size_t)to];
	for (int side = 0; side < 2; ++side)
	{
		const int g = d.seamGroup[side];
		if (g < 0)
		{
			continue;
		}
		const int n = m.seamBodyCount[(size_t)g]; // (the bodies the seam has come to carry since the build included)
		for (int e = d.remapBase[side]; e < d.remapBase[side] + n; ++e)
		{
			if (s->hPersistRemap[(size_t)e] >= nb)
			{
				s->hPersistRemap[(size_t)e] += 1;
				p.word(s->persist.remap, (size_t)e, (uint32_t)s->hPersistRemap[(size_t)e]);
			}
		}
	}
	m.adopted += 1;
	m.touched = true;
	static const bool debug = getenv("S2AMD_DEBUG_PLACE") != nullptr;
	if (debug)
	{
		fprintf(stderr, "[s2amd] body %d moves from strip %d (slot %d) to strip %d (slot %d, list at %d)\n", body, from, fromSlot, to, nb, m.stripBodyBase[(size_t)to]);
	}
}

// `body`, owned by one of the two strips of seam `sm` (seam group g), becomes the seam's next local body: IncrementalStrips says what that takes
static void seamExtend(s2amdSolver* s, Patcher& p, int sm, int g, int body)
{
	IncrementalStrips& m = s->stripInc;
	const int sx = m.ownerStrip[(size_t)body], so = sx == sm ? sm + 1 : sm;
	const int sideX = sx == sm ? 1 : 0, sideO = 1 - sideX; // seam sm is side 1 of strip sm and side 0 of strip sm + 1
	PersistDesc& dx = s->hPersistDescs[(size_t)sx];
	PersistDesc& dO = s->hPersistDescs[(size_t)so];
	const int slot = m.seamBodyCount[(size_t)g];
	m.seamBodyCount[(size_t)g] = slot + 1;
	m.seamExtra[sx == sm ? 0 : 1][(size_t)g] += 1;
	auto descWord = [&](int strip, size_t byteOffset, uint32_t value) {
		const size_t words = (size_t)((const uint32_t*)(s->persist.descs + strip) - (const uint32_t*)s->persist.descs);
		p.word(s->persist.descs, words + byteOffset / 4, value);
	};
	auto setRemap = [&](int e, int value) {
		s->hPersistRemap[(size_t)e] = value;
		p.word(s->persist.remap, (size_t)e, (uint32_t)value);
	};
	// the owner exports it ...
	const int index = dx.exportCount[sideX]; // (== the neighbour's import count on this seam: the two lists are one list)
	p.word(s->persist.exportSrc, (size_t)(dx.exportSrcBase[sideX] + index), (uint32_t)m.ownerSlot[(size_t)body]);
	dx.exportCount[sideX] = index + 1;
	descWord(sx, offsetof(PersistDesc, exportCount) + 4 * (size_t)sideX, (uint32_t)(index + 1));
	setRemap(dx.remapBase[sideX] + slot, m.ownerSlot[(size_t)body]);
	// ... the neighbour imports it: behind its own bodies and (side 1) the left seam's imports; a new import of the LEFT seam moves the
	// right seam's imports one LDS slot on
	const int nbO = m.stripBodyCount[(size_t)so];
	const int ldsSlot = nbO + (sideO == 1 ? dO.importCount[0] : 0) + index;
	p.word(s->persist.importIds, (size_t)(dO.importIdBase[sideO] + index), (uint32_t)body);
	if (sideO == 0 && dO.seamGroup[1] >= 0)
	{
		const int n1 = m.seamBodyCount[(size_t)dO.seamGroup[1]];
		for (int e = dO.remapBase[1]; e < dO.remapBase[1] + n1; ++e)
		{
			if (s->hPersistRemap[(size_t)e] >= nbO + index)
			{
				setRemap(e, s->hPersistRemap[(size_t)e] + 1);
			}
		}
	}
	dO.importCount[sideO] = index + 1;
	descWord(so, offsetof(PersistDesc, importCount) + 4 * (size_t)sideO, (uint32_t)(index + 1));
	setRemap(dO.remapBase[sideO] + slot, ldsSlot);
	m.adoptedBy[(size_t)so] += 1; // (its LDS budget: one more staged body)
	// the host's picture: the seam's local slot, its round mask (the group's entries move behind the others when they outgrow their place)
	m.seamSlot[(size_t)g][body] = slot;
	const int off = m.bodyOffset[1][(size_t)g];
	const int built = s->hStripB.bodyOffsets[(size_t)g + 1] - s->hStripB.bodyOffsets[(size_t)g];
	if (slot == built) // (the first appended body: as built, the next group's entries follow directly)
	{
		const int newOff = (int)m.roundMask[1].size();
		m.roundMask[1].resize((size_t)(newOff + built + 2 * S2_STRIP_ADOPT_SLACK), 0u);
		std::copy(m.roundMask[1].begin() + off, m.roundMask[1].begin() + off + built, m.roundMask[1].begin() + newOff);
		m.bodyOffset[1][(size_t)g] = newOff;
	}
	m.roundMask[1][(size_t)(m.bodyOffset[1][(size_t)g] + slot)] = 0u;
	m.seamBodiesAdded += 1;
	m.touched = true;
	static const bool debug = getenv("S2AMD_DEBUG_PLACE") != nullptr;
	if (debug)
	{
		fprintf(stderr, "[s2amd] seam %d (group %d) now carries body %d of strip %d: local %d, export %d on side %d (own slot %d), import of strip %d on side %d at LDS %d (its own bodies %d, imports %d + %d)\n", sm, g,
				body, sx, slot, index, sideX, m.ownerSlot[(size_t)body], so, sideO, ldsSlot, nbO, dO.importCount[0], dO.importCount[1]);
	}
}

bool stripPlace(s2amdSolver* s, Patcher& p, const ContactChange& ch)
{
	IncrementalStrips& m = s->stripInc;
	int table, group, la, lb, mover = -1;
	if (!m.valid || ch.slot >= (int)m.positionOfSlot.size() || m.positionOfSlot[(size_t)ch.slot] >= 0 || !stripHome(s, ch.a, ch.b, table, group, la, lb, &mover))
	{
		return false;
	}
	if (mover >= 0)
	{
		stripAdopt(s, p, mover, group); // (its round mask is empty: the search below finds the strip's first round with a free position)
	}
	else if (mover <= -3)
	{
		const int missing = -2 - mover;
		const int sm = std::min(m.ownerStrip[(size_t)ch.a], m.ownerStrip[(size_t)ch.b]);
		if (missing & 1)
		{
			seamExtend(s, p, sm, group, ch.a);
		}
		if (missing & 2)
		{
			seamExtend(s, p, sm, group, ch.b);
		}
	}
	const int off = m.bodyOffset[table][(size_t)group];
	const bool wa = writable(s, ch.a), wb = writable(s, ch.b);
	const uint32_t used = (wa ? m.roundMask[table][(size_t)(off + la)] : 0u) | (wb ? m.roundMask[table][(size_t)(off + lb)] : 0u);
	auto take = [&](IncrementalStrips::Round& round, int r) {
		const int k = round.freePositions.back();
		round.freePositions.pop_back();
		if (wa)
		{
			m.roundMask[table][(size_t)(off + la)] |= 1u << r;
		}
		if (wb)
		{
			m.roundMask[table][(size_t)(off + lb)] |= 1u << r;
		}
		s->contacts.order[(size_t)k] = ch.slot;
		s->contacts.local[(size_t)k] = make_int2(la, lb);
		m.positionOfSlot[(size_t)ch.slot] = k;
		s->inc.positionOfSlot[(size_t)ch.slot] = -2;
		p.word(s->dContactIndex.p, (size_t)k, (uint32_t)ch.slot);
		p.word(s->dContactLocal.p, 2 * (size_t)k, (uint32_t)la);
		p.word(s->dContactLocal.p, 2 * (size_t)k + 1, (uint32_t)lb);
		m.touched = true;
		m.placed += 1;
		s->placedTotal += 1;
		s->slackPositions -= 1;
	};
	std::vector<int>& open = m.roundsOf[table][(size_t)group];
	const int n = (int)open.size();
	for (int r = 0; r < n; ++r)
	{
		IncrementalStrips::Round& round = m.rounds[(size_t)open[(size_t)r]];
		if (((used >> r) & 1u) != 0 || round.freePositions.empty())
		{
			continue;
		}
		take(round, r);
		return true;
	}
	std::vector<int>& spare = m.spareOf[table][(size_t)group];
	if (!spare.empty() && s->optStripAdopt != 0 && (int)s->hPersistDescs.size() == (int)m.stripBodyCount.size())
	{
		// every round is taken on these bodies: the group's next spare round opens -- the round count and the new round's range in
		// the strip's descriptor (a seam: in the descriptors of its two strips) -- and takes the constraint.  Beyond six interior
		// rounds or two seam rounds the step needs another variant of the kernel: the launch picks it from pv.maxRoundsA /
		// maxSeamRounds, a captured step graph is dropped (layoutGeneration).
		IncrementalStrips::Round& round = m.rounds[(size_t)spare.front()];
		const int r = round.round;
		const int begin = round.freePositions.back(), end = round.freePositions.front() + 1;
		if (r != n || r >= 32 || r + 1 > m.roundLimit[table])
		{
			return false; // (... or one round more than this solver's kernel has a layout for: the overflow positions, or a rebuild)
		}
		if (table == 0)
		{
			static_assert(offsetof(StripDesc, batchCount) % 4 == 0 && offsetof(StripDesc, batch) % 4 == 0 && sizeof(StripDesc::batch[0]) == 16, "StripDesc layout");
			if (r + 1 > S2_STRIP_ROUNDS_MAX)
			{
				return false;
			}
			const StripDesc* desc = s->leanA.descs + group;
			const size_t words = (const uint32_t*)desc - (const uint32_t*)s->leanA.descs;
			const size_t wCount = offsetof(StripDesc, batchCount) / 4, wBatch = offsetof(StripDesc, batch) / 4;
			p.word(s->leanA.descs, words + wCount, (uint32_t)(r + 1));
			p.word(s->leanA.descs, words + wBatch + 4 * (size_t)r, (uint32_t)begin); // batch[r] = {begin, end, 0, 0}
			p.word(s->leanA.descs, words + wBatch + 4 * (size_t)r + 1, (uint32_t)end);
			if (r + 1 > s->persist.maxRoundsA)
			{
				s->persist.maxRoundsA = r + 1;
				if (r + 1 > S2_STRIP_ROUNDS)
				{
					s->persist.wideOnly = 1; // (the 256-thread kernels' variant flags and LDS budgets are as built: they stay off these strips)
					s->layoutGeneration += 1;
				}
			}
		}
		else
		{
			static_assert(offsetof(PersistDesc, seamBatchCount) % 4 == 0 && offsetof(PersistDesc, seamBatch) % 4 == 0 && sizeof(int2) == 8, "PersistDesc layout");
			const int sm = group < (int)m.seamOfGroup.size() ? m.seamOfGroup[(size_t)group] : -1;
			if (r + 1 > S2_PERSIST_B_ROUNDS || sm < 0)
			{
				return false;
			}
			for (int side = 0; side < 2; ++side)
			{
				const int strip = side == 1 ? sm : sm + 1; // seam sm is side 1 of strip sm and side 0 of strip sm + 1
				PersistDesc& d = s->hPersistDescs[(size_t)strip];
				d.seamBatchCount[side] = r + 1;
				d.seamBatch[side][r] = make_int2(begin, end);
				const size_t words = (size_t)((const uint32_t*)(s->persist.descs + strip) - (const uint32_t*)s->persist.descs);
				p.word(s->persist.descs, words + offsetof(PersistDesc, seamBatchCount) / 4 + (size_t)side, (uint32_t)(r + 1));
				const size_t wBatch = offsetof(PersistDesc, seamBatch) / 4 + 2 * ((size_t)side * S2_PERSIST_B_ROUNDS + (size_t)r);
				p.word(s->persist.descs, words + wBatch, (uint32_t)begin);
				p.word(s->persist.descs, words + wBatch + 1, (uint32_t)end);
			}
			if (r + 1 > s->persist.maxSeamRounds)
			{
				s->persist.maxSeamRounds = r + 1;
				if (r + 1 > 2)
				{
					s->persist.wideOnly = 1;
					s->layoutGeneration += 1;
				}
			}
		}
		open.push_back(spare.front());
		spare.erase(spare.begin());
		m.roundsOpened += 1;
		take(round, r);
		return true;
	}
	static const bool debug = getenv("S2AMD_DEBUG_PLACE") != nullptr;
	if (debug)
	{
		fprintf(stderr, "[s2amd] no free round for (%d, %d) in table %d group %d: rounds %d, used mask %x\n", ch.a, ch.b, table, group, n, used);
	}
	return false; // every round is taken on these bodies, or full
}

// A contact between bodies the strips own that fits in none of their rounds: the lowest free overflow position (solver_internal.h:
// IncrementalStrips).  One patched word -- contactIndex[k]; the colour-batch kernels that sweep it read the bodies' pool slots, which
// the prologue writes for every position (contact_kernels.hip: prepareContactsKernel).
bool overflowPlace(s2amdSolver* s, Patcher& p, const ContactChange& ch)
{
	IncrementalStrips& m = s->stripInc;
	if (!overflowCanPlace(s, ch.a, ch.b) || ch.slot >= (int)m.positionOfSlot.size() || m.positionOfSlot[(size_t)ch.slot] >= 0)
	{
		return false;
	}
	// the bodies' entries in the list the overflow workgroup stages (wide_kernel.hip: wideOverflowWorker; the sliced launches do not
	// read it): found or made -- when the list is full the contact cannot wait here
	const uint32_t wbit = s->inc.solverClass == 1 ? S2F_WRITE_POS : S2F_WRITE_VEL;
	int entry[2] = {-1, -1};
	{
		std::vector<int> ids = m.overflowBodyIds;
		const int bodies[2] = {ch.a, ch.b};
		for (int side = 0; side < 2; ++side)
		{
			const int code = bodies[side] | ((s->hBodyFlags[(size_t)bodies[side]] & wbit) != 0 ? 0 : 0x40000000);
			int at = -1, hole = -1;
			for (int i = 0; i < (int)ids.size(); ++i)
			{
				at = ids[(size_t)i] == code ? i : at;
				hole = (ids[(size_t)i] < 0 && hole < 0) ? i : hole;
			}
			if (at < 0 && hole < 0 && (int)ids.size() < S2_OVERFLOW_BODIES)
			{
				hole = (int)ids.size();
				ids.push_back(-1);
			}
			if (at < 0 && hole < 0)
			{
				return false;
			}
			if (at < 0)
			{
				ids[(size_t)hole] = code;
				at = hole;
			}
			entry[side] = at;
		}
		for (int i = 0; i < (int)ids.size(); ++i)
		{
			if (i >= (int)m.overflowBodyIds.size() || m.overflowBodyIds[(size_t)i] != ids[(size_t)i])
			{
				p.word(s->dOverflowBodies.p, (size_t)i, (uint32_t)ids[(size_t)i]);
			}
		}
		m.overflowBodyIds = std::move(ids);
	}
	const int k = m.overflowFree.back();
	m.overflowFree.pop_back();
	s->contacts.order[(size_t)k] = ch.slot;
	s->contacts.local[(size_t)k] = make_int2(entry[0], entry[1]);
	m.positionOfSlot[(size_t)ch.slot] = k;
	s->inc.positionOfSlot[(size_t)ch.slot] = -2;
	p.word(s->dContactIndex.p, (size_t)k, (uint32_t)ch.slot);
	p.word(s->dContactLocal.p, 2 * (size_t)k, (uint32_t)entry[0]);
	p.word(s->dContactLocal.p, 2 * (size_t)k + 1, (uint32_t)entry[1]);
	m.overflowUsed += 1;
	m.overflowPlaced += 1;
	s->placedTotal += 1;
	s->slackPositions -= 1;
	s->layoutGeneration += 1; // (the launch sequence changes: a captured step graph is dropped)
	return true;
}

// the entry of `slot` in the strips becomes a free position again
bool stripRemove(s2amdSolver* s, Patcher& p, int slot)
{
	IncrementalStrips& m = s->stripInc;
	if (!m.valid || slot >= (int)m.positionOfSlot.size() || m.positionOfSlot[(size_t)slot] < 0)
	{
		return false;
	}
	const int k = m.positionOfSlot[(size_t)slot];
	if (k >= m.overflowBegin && k < m.overflowEnd)
	{
		// an overflow position (no round, no local slots): free again; the launch sequence loses its sweeps
		s->contacts.order[(size_t)k] = -1;
		p.word(s->dContactIndex.p, (size_t)k, (uint32_t)-1);
		{
			// entries of the overflow workgroup's body list that no other overflow contact refers to become free
			const int2 mine = s->contacts.local[(size_t)k];
			bool used[2] = {false, false};
			for (int o = m.overflowBegin; o < m.overflowEnd; ++o)
			{
				if (o != k && s->contacts.order[(size_t)o] >= 0)
				{
					const int2 l = s->contacts.local[(size_t)o];
					used[0] = used[0] || l.x == mine.x || l.y == mine.x;
					used[1] = used[1] || l.x == mine.y || l.y == mine.y;
				}
			}
			const int e[2] = {mine.x, mine.y};
			for (int side = 0; side < 2; ++side)
			{
				if (!used[side] && e[side] >= 0 && e[side] < (int)m.overflowBodyIds.size() && m.overflowBodyIds[(size_t)e[side]] >= 0)
				{
					m.overflowBodyIds[(size_t)e[side]] = -1;
					p.word(s->dOverflowBodies.p, (size_t)e[side], (uint32_t)-1);
				}
			}
			s->contacts.local[(size_t)k] = make_int2(0, 0);
		}
		m.overflowFree.insert(std::upper_bound(m.overflowFree.begin(), m.overflowFree.end(), k, std::greater<int>()), k); // stays descending
		m.overflowUsed -= 1;
		m.positionOfSlot[(size_t)slot] = -1;
		s->inc.positionOfSlot[(size_t)slot] = -1;
		s->slackPositions += 1;
		s->layoutGeneration += 1;
		return true;
	}
	IncrementalStrips::Round& round = m.rounds[(size_t)m.roundOfPosition[(size_t)(k - m.base)]];
	const int off = m.bodyOffset[round.table][(size_t)round.group];
	const int2 l = s->contacts.local[(size_t)k];
	const int oa = s->hContactA[(size_t)slot], ob = s->hContactB[(size_t)slot]; // the endpoints the structure knows
	if (oa >= 0 && oa < (int)s->hBodyFlags.size() && writable(s, oa))
	{
		m.roundMask[round.table][(size_t)(off + l.x)] &= ~(1u << round.round);
	}
	if (ob >= 0 && ob < (int)s->hBodyFlags.size() && writable(s, ob))
	{
		m.roundMask[round.table][(size_t)(off + l.y)] &= ~(1u << round.round);
	}
	s->contacts.order[(size_t)k] = -1;
	s->contacts.local[(size_t)k] = make_int2(0, 0);
	p.word(s->dContactIndex.p, (size_t)k, (uint32_t)-1);
	p.word(s->dContactLocal.p, 2 * (size_t)k, 0u);
	p.word(s->dContactLocal.p, 2 * (size_t)k + 1, 0u);
	std::vector<int>& fp = round.freePositions;
	fp.insert(std::upper_bound(fp.begin(), fp.end(), k, std::greater<int>()), k); // stays descending
	m.positionOfSlot[(size_t)slot] = -1;
	s->inc.positionOfSlot[(size_t)slot] = -1;
	m.touched = true;
	s->slackPositions += 1;
	return true;
}

} // namespace

// the entry of `slot` leaves the structure; false: it cannot (not in a parallel batch of the global part)
static bool removeEntry(s2amdSolver* s, Patcher& p, int slot)
{
	IncrementalGlobal& inc = s->inc;
	const int W = 4;
	const int kOld = inc.positionOfSlot[(size_t)slot];
	if (kOld == -2 && stripRemove(s, p, slot))
	{
		return true;
	}
	if (kOld < 0)
	{
		return kOld == -1; // -1: no entry (nothing to do); -2: it lives in an LDS group or a strip
	}
	const int bi = inc.colorOfPosition[(size_t)kOld];
	if (bi < 0)
	{
		// in the sequential tail: a free position again when the tail is laid out with slack (IncrementalGlobal::tailFree)
		if (kOld < inc.tailBegin || kOld >= inc.tailEnd || inc.tailBodyCapacity == 0)
		{
			return false;
		}
		const int oa = s->hContactA[(size_t)slot], ob = s->hContactB[(size_t)slot];
		for (int side = 0; side < 2; ++side)
		{
			const int body = side == 0 ? oa : ob;
			if (body >= 0 && body < (int)s->hBodyFlags.size() && writable(s, body))
			{
				adjRemove(p, body, (kOld << 1) | side);
			}
		}
		s->contacts.order[(size_t)kOld] = -1;
		s->contacts.local[(size_t)kOld] = make_int2(0, 0);
		p.word(s->dContactIndex.p, (size_t)kOld, (uint32_t)-1);
		p.word(s->dContactLocal.p, 2 * (size_t)kOld, 0u);
		p.word(s->dContactLocal.p, 2 * (size_t)kOld + 1, 0u);
		inc.tailFree.insert(std::upper_bound(inc.tailFree.begin(), inc.tailFree.end(), kOld, std::greater<int>()), kOld); // stays descending
		inc.positionOfSlot[(size_t)slot] = -1;
		inc.removed += 1;
		s->slackPositions += 1;
		return true;
	}
	const int cid = inc.colorIdOfBatch[(size_t)bi];
	const int oa = s->hContactA[(size_t)slot], ob = s->hContactB[(size_t)slot]; // the endpoints the structure knows
	for (int side = 0; side < 2; ++side)
	{
		const int body = side == 0 ? oa : ob;
		if (body >= 0 && body < (int)s->hBodyFlags.size() && writable(s, body))
		{
			if (cid < 64 * W && !inc.ignoreColours)
			{
				inc.colorBits[(size_t)body * W + (size_t)(cid / 64)] &= ~(1ull << (cid % 64));
			}
			adjRemove(p, body, (kOld << 1) | side);
		}
	}
	s->contacts.order[(size_t)kOld] = -1;
	p.word(s->dContactIndex.p, (size_t)kOld, (uint32_t)-1);
	std::vector<int>& fp = inc.freePositions[(size_t)bi];
	fp.insert(std::upper_bound(fp.begin(), fp.end(), kOld, std::greater<int>()), kOld); // stays descending
	inc.positionOfSlot[(size_t)slot] = -1;
	inc.removed += 1;
	s->slackPositions += 1;
	return true;
}

void incrementalRemove(s2amdSolver* s, const int32_t* slots, int count)
{
	IncrementalGlobal& inc = s->inc;
	if (!inc.valid || s->structureDirty || s->optIncremental == 0)
	{
		return;
	}
	Patcher p{s, inc};
	for (int i = 0; i < count; ++i)
	{
		const int slot = slots[i];
		if (slot >= 0 && slot < (int)inc.positionOfSlot.size() && s->hContactEdge[(size_t)slot] && removeEntry(s, p, slot))
		{
			s->hContactEdge[(size_t)slot] = 0; // gone from the structure (else: it lingers, hContactDead says so)
			s->hContactDead[(size_t)slot] = 0;
		}
	}
}

bool incrementalApply(s2amdSolver* s, const std::vector<ContactChange>& changes)
{
	IncrementalGlobal& inc = s->inc;
	if (!inc.valid || s->structureDirty || s->optIncremental == 0)
	{
		return false;
	}
	auto giveUp = [&](const char* why = "placement") {
		s->dirtyReason = why;
		// nothing has reached the device; the host mirrors are rebuilt with the structure
		inc.valid = false;
		inc.patches.clear();
		inc.fallbacks += 1;
		return false;
	};
	Patcher p{s, inc};
	SweepSet& cs = s->contacts;
	const int W = 4;
	inc.placedInGlobalPart = false;
	for (const ContactChange& ch : changes)
	{
		if (ch.slot < 0 || ch.slot >= (int)inc.positionOfSlot.size())
		{
			return giveUp("slot outside the structure");
		}
		// ---- the slot's previous entry (a destroyed contact whose entry lingered) gives its place back ----
		if (!removeEntry(s, p, ch.slot))
		{
			// (an entry in an LDS group's tables, unless strips hold it: SolverRest::groupPatienceNow)
			s->dirtyByGroups = inc.positionOfSlot[(size_t)ch.slot] == -2 && (ownedByLdsGroup(s, s->hContactA[(size_t)ch.slot]) || ownedByLdsGroup(s, s->hContactB[(size_t)ch.slot]));
			return giveUp("old entry of the slot not removable");
		}
		if (ch.a < 0)
		{
			continue;
		}
		// ---- the new contact ----
		const int nb = (int)s->hBodyFlags.size();
		if (ch.a >= nb || ch.b < 0 || ch.b >= nb || ch.a == ch.b || s->hBodyFlagsFinal.size() != (size_t)nb)
		{
			return giveUp("bad body");
		}
		if ((s->hBodyFlagsFinal[(size_t)ch.a] & S2F_IN_GROUP) != 0 || (s->hBodyFlagsFinal[(size_t)ch.b] & S2F_IN_GROUP) != 0)
		{
			// a strip owns a body: a free position of one of its rounds (IncrementalStrips); an LDS group's tables are not placeable into
			if (stripPlace(s, p, ch))
			{
				inc.inserted += 1;
				unwatchSlot(s, ch.slot); // structural from here on: its manifold may gain and lose points at no cost
				continue;
			}
			if (overflowPlace(s, p, ch))
			{
				inc.inserted += 1;
				unwatchSlot(s, ch.slot);
				continue;
			}
			s->dirtyByGroups = ownedByLdsGroup(s, ch.a) || ownedByLdsGroup(s, ch.b);
			return giveUp("body owned by a group or strip");
		}
		inc.placedInGlobalPart = true;
		const bool wa = writable(s, ch.a), wb = writable(s, ch.b);
		int chosen = -1;
		for (int bi = 0; bi < inc.parallelBatches; ++bi)
		{
			if (inc.freePositions[(size_t)bi].empty())
			{
				continue;
			}
			const int cid = inc.colorIdOfBatch[(size_t)bi];
			if (inc.ignoreColours)
			{
				chosen = bi;
				inc.colourFreePlaced = true;
				s->jacobiValid = false; // (the persistent launch's block tables know nothing of this constraint: the multi-launch path until the next build)
				break;
			}
			if (cid >= 64 * W)
			{
				continue;
			}
			const uint64_t used = (wa ? inc.colorBits[(size_t)ch.a * W + (size_t)(cid / 64)] : 0) | (wb ? inc.colorBits[(size_t)ch.b * W + (size_t)(cid / 64)] : 0);
			if (((used >> (cid % 64)) & 1ull) == 0)
			{
				chosen = bi;
				break;
			}
		}
		if (chosen < 0 && !inc.ignoreColours && !inc.tailFree.empty())
		{
			// no parallel colour is free on these bodies (a contact of a hub: it uses every one): the sequential tail takes it -- the
			// lowest free position behind its constraints; a body the tail does not stage yet joins its body list
			int local[2] = {0, 0};
			int fresh = 0;
			for (int side = 0; side < 2; ++side)
			{
				const int body = side == 0 ? ch.a : ch.b;
				const auto it = inc.tailBodySlot.find(body);
				local[side] = it != inc.tailBodySlot.end() ? it->second : inc.tailBodyCount + fresh++;
			}
			if (inc.tailBodyCount + fresh <= inc.tailBodyCapacity && s->dContactTail.view.groupCount == 1)
			{
				for (int side = 0; side < 2; ++side)
				{
					const int body = side == 0 ? ch.a : ch.b;
					if (inc.tailBodySlot.find(body) == inc.tailBodySlot.end())
					{
						const bool w = side == 0 ? wa : wb;
						inc.tailBodySlot[body] = local[side];
						p.word(s->dContactTail.view.bodyIds, (size_t)local[side], (uint32_t)body | (w ? S2G_OWNED : 0u));
					}
				}
				if (fresh > 0)
				{
					inc.tailBodyCount += fresh;
					p.word(s->dContactTail.view.bodyOffsets, 1, (uint32_t)inc.tailBodyCount);
				}
				const int k = inc.tailFree.back();
				inc.tailFree.pop_back();
				cs.order[(size_t)k] = ch.slot;
				cs.local[(size_t)k] = make_int2(local[0], local[1]);
				inc.positionOfSlot[(size_t)ch.slot] = k;
				p.word(s->dContactIndex.p, (size_t)k, (uint32_t)ch.slot);
				p.word(s->dContactLocal.p, 2 * (size_t)k, (uint32_t)local[0]);
				p.word(s->dContactLocal.p, 2 * (size_t)k + 1, (uint32_t)local[1]);
				if ((wa && !adjInsert(p, ch.a, (k << 1) | 0)) || (wb && !adjInsert(p, ch.b, (k << 1) | 1)))
				{
					s->slackShift = std::min(s->slackShift + 1, 3);
					s->slackBumped = true;
					return giveUp(inc.adjFailure == 2 ? "list of heavy bodies full (tail)" : "incidence list full");
				}
				inc.inserted += 1;
				inc.tailPlaced += 1;
				s->placedTotal += 1;
				s->slackPositions -= 1;
				unwatchSlot(s, ch.slot);
				continue;
			}
		}
		if (chosen < 0)
		{
			bool anyFree = false;
			for (int bi = 0; bi < inc.parallelBatches; ++bi)
			{
				anyFree = anyFree || !inc.freePositions[(size_t)bi].empty();
			}
			if (!anyFree || inc.ignoreColours)
			{
				// the slack itself is used up (a world that creates contacts by the thousand per step): the rebuild lays out more
				s->slackShift = std::min(s->slackShift + 1, 3);
				s->slackBumped = true;
				return giveUp("no free position");
			}
			// every colour batch is taken on these bodies: the rebuild adds empty ones for the next such contact.  (Twice as many
			// every time -- up to 16 -- was measured on the Tumbler filled from scratch, r4: fewer rebuilds, 57 instead of 67 of 120
			// steps, but every spare colour in use is a launch per sweep: 287 instead of 257 launches per step, no faster.)
			// (r6: ... but where the structures die of it one after the other -- a pile coming down under the reference's default solver:
			// 98 of 105 builds, one per step -- twice as many each time: an empty colour costs nothing until it is used)
			s->spareColours = s->graphAge < 8 ? std::min(std::max(2 * s->spareColours, 2), 16) : std::max(s->spareColours, 2);
			return giveUp("no free colour");
		}
		if ((int)inc.freePositions[(size_t)chosen].size() == inc.batchEnd[(size_t)chosen] - inc.batchBegin[(size_t)chosen])
		{
			s->layoutGeneration += 1; // the batch was empty: its launches were left out of the captured step graph (Executor::emptyBatch)
		}
		const int k = inc.freePositions[(size_t)chosen].back();
		inc.freePositions[(size_t)chosen].pop_back();
		if (!inc.ignoreColours)
		{
			const int cid = inc.colorIdOfBatch[(size_t)chosen];
			if (wa)
			{
				inc.colorBits[(size_t)ch.a * W + (size_t)(cid / 64)] |= 1ull << (cid % 64);
			}
			if (wb)
			{
				inc.colorBits[(size_t)ch.b * W + (size_t)(cid / 64)] |= 1ull << (cid % 64);
			}
		}
		cs.order[(size_t)k] = ch.slot;
		inc.positionOfSlot[(size_t)ch.slot] = k;
		p.word(s->dContactIndex.p, (size_t)k, (uint32_t)ch.slot);
		if ((wa && !adjInsert(p, ch.a, (k << 1) | 0)) || (wb && !adjInsert(p, ch.b, (k << 1) | 1)))
		{
			s->slackShift = std::min(s->slackShift + 1, 3);
			s->slackBumped = true;
			return giveUp(inc.adjFailure == 2 ? "list of heavy bodies full" : "incidence lists full");
		}
		inc.inserted += 1;
		s->placedTotal += 1;
		s->slackPositions -= 1;
	}
	return true;
}

int incrementalFlush(s2amdSolver* s)
{
	IncrementalGlobal& inc = s->inc;
	if (inc.patches.empty())
	{
		return S2AMD_OK;
	}
	{
		// one thread applies one patch: a word that was written more than once (a list entry that moved and was then shifted,
		// a range that two insertions touched) must appear once, with its LAST value
		std::unordered_map<unsigned long long, size_t> last;
		last.reserve(inc.patches.size() * 2);
		std::vector<uint4> unique;
		unique.reserve(inc.patches.size());
		for (const uint4& q : inc.patches)
		{
			const unsigned long long addr = ((unsigned long long)q.y << 32) | q.x;
			auto it = last.find(addr);
			if (it == last.end())
			{
				last.emplace(addr, unique.size());
				unique.push_back(q);
			}
			else
			{
				unique[it->second].z = q.z;
			}
		}
		inc.patches.swap(unique);
	}
	const size_t n = inc.patches.size();
	HIP_TRY(hipSetDevice(s->device));
	if (n > s->hostPatchCapacity)
	{
		// the staging buffer may still be read by a copy enqueued earlier
		HIP_TRY(hipStreamSynchronize(s->stream));
		if (s->hostPatches)
		{
			(void)hipHostFree(s->hostPatches);
			s->hostPatches = nullptr;
		}
		size_t cap = std::max<size_t>(2 * n, 4096);
		size_t got = 0;
		void* pooled = devPoolOn() ? pinnedPoolTake(cap * sizeof(uint4), &got) : nullptr; // (a worker's copy: solver_internal.h)
		if (pooled)
		{
			s->hostPatches = (uint4*)pooled;
			cap = got / sizeof(uint4);
		}
		else
		{
			HIP_TRY(hipHostMalloc((void**)&s->hostPatches, cap * sizeof(uint4), hipHostMallocDefault));
		}
		s->hostPatchCapacity = cap;
	}
	else
	{
		HIP_TRY(hipStreamSynchronize(s->stream)); // (callers are between steps: the stream is idle; this makes the reuse of the staging buffer safe)
	}
	int rc = s->dPatches.ensure(std::max<size_t>(s->hostPatchCapacity, n) * sizeof(uint4));
	if (rc)
	{
		return rc;
	}
	memcpy(s->hostPatches, inc.patches.data(), n * sizeof(uint4));
	HIP_TRY(hipMemcpyAsync(s->dPatches.p, s->hostPatches, n * sizeof(uint4), hipMemcpyHostToDevice, s->stream));
	launchPatchWords(s->stream, s->dPatches.p, (int)n);
	HIP_TRY(hipGetLastError());
	inc.patches.clear();
	return S2AMD_OK;
}

// ---- deferred contacts (solver_internal.h: optDefer) ----
bool canDeferCreated(const s2amdSolver* s, int slot, int a, int b)
{
	const int nb = (int)s->hBodyFlagsFinal.size();
	if (s->optDefer == 0 || s->structureDirty || slot < 0 || slot >= (int)s->inc.positionOfSlot.size() || a < 0 || b < 0 || a >= nb || b >= nb || a == b ||
		(int)s->hContactWatched.size() != s->contactCapacity)
	{
		return false;
	}
	if (s->inc.positionOfSlot[(size_t)slot] != -1)
	{
		return false; // the slot still has an entry (a destroyed contact's, lingering): that one has to go first
	}
	const bool owned = ((s->hBodyFlagsFinal[(size_t)a] | s->hBodyFlagsFinal[(size_t)b]) & S2F_IN_GROUP) != 0;
	const bool hub = (int)s->hBodyHub.size() == nb && (s->hBodyHub[(size_t)a] || s->hBodyHub[(size_t)b]);
	return owned || hub;
}

// a watched manifold between these bodies got its first points: can it take a place in the strips (instead of a rebuild)?
bool ownedByLdsGroup(const s2amdSolver* s, int body)
{
	return body >= 0 && body < (int)s->hBodyLdsOwned.size() && s->hBodyLdsOwned[(size_t)body] != 0;
}

bool stripCanPlace(const s2amdSolver* s, int a, int b)
{
	const int nb = (int)s->hBodyFlagsFinal.size();
	if (!s->stripInc.valid || s->structureDirty || a < 0 || b < 0 || a >= nb || b >= nb || a == b)
	{
		return false;
	}
	if ((int)s->hBodyHub.size() == nb && (s->hBodyHub[(size_t)a] || s->hBodyHub[(size_t)b]))
	{
		return false;
	}
	int table, group, la, lb, mover = -1;
	return stripHome(s, a, b, table, group, la, lb, &mover);
}

bool overflowCanPlace(const s2amdSolver* s, int a, int b)
{
	const IncrementalStrips& m = s->stripInc;
	const int nb = (int)s->hBodyFlagsFinal.size();
	if (!m.valid || m.overflowFree.empty() || s->optOverflow == 0 || s->optIncremental == 0 || s->structureDirty || a < 0 || b < 0 || a >= nb || b >= nb || a == b)
	{
		return false;
	}
	// the sliced step is the 512-thread persistent kernel's (wide_kernel.hip), and a structure that will hold the contact has to come
	// from a worker thread while the steps go on: the world chain
	// (a worker's copy takes overflow contacts too -- the changes replayed on it at its adoption: the structure that replaces the live one
	// may itself run sliced until the next one is adopted)
	if (!(asyncBuildsOn(s) || (s->isClone && s->optAsyncBuild != 0)) || !s->persistValid || s->persistFailed || s->optWide == 0 || s->optPersist == 0)
	{
		return false;
	}
	if ((int)s->hBodyHub.size() == nb && (s->hBodyHub[(size_t)a] || s->hBodyHub[(size_t)b]))
	{
		return false;
	}
	// both bodies live in the strips (owned by one, or read-only replicas of bodies nothing writes)
	auto inStrips = [&](int body) {
		const bool writes = (s->hBodyFlags[(size_t)body] & (s->inc.solverClass == 1 ? S2F_WRITE_POS : S2F_WRITE_VEL)) != 0;
		return !writes || (body < (int)m.ownerStrip.size() && m.ownerStrip[(size_t)body] >= 0);
	};
	return inStrips(a) && inStrips(b);
}

bool tailCanPlace(const s2amdSolver* s, int a, int b)
{
	const int nb = (int)s->hBodyFlagsFinal.size();
	// (r6: with or without a tail -- a pile of boxes pressed together has hundreds of bodies with more than S2_HUB_DEGREE potential
	// contacts and no colour small enough to make a tail; their manifolds take a free colour position like any created contact, and only
	// the one that finds none -- incrementalApply -- costs the build that every one of them used to cost: 134 of 200 wreck steps under XPBD)
	if (!s->inc.valid || s->inc.ignoreColours || (s->optFlipColours == 0 && s->inc.tailFree.empty()) || s->optIncremental == 0 || s->structureDirty || a < 0 || b < 0 ||
		a >= nb || b >= nb || a == b)
	{
		return false;
	}
	return ((s->hBodyFlagsFinal[(size_t)a] | s->hBodyFlagsFinal[(size_t)b]) & S2F_IN_GROUP) == 0;
}

void deferCreated(s2amdSolver* s, int slot, int a, int b)
{
	s->hContactA[(size_t)slot] = a;
	s->hContactB[(size_t)slot] = b;
	s->hContactEdge[(size_t)slot] = 1;
	s->hContactDead[(size_t)slot] = 0;
	if (!s->hContactWatched[(size_t)slot])
	{
		s->hContactWatched[(size_t)slot] = 1;
		s->watchedCount += 1;
		s->watchedDirty = true;
	}
}

void unwatchSlot(s2amdSolver* s, int slot)
{
	if (slot >= 0 && slot < (int)s->hContactWatched.size() && s->hContactWatched[(size_t)slot])
	{
		s->hContactWatched[(size_t)slot] = 0;
		s->watchedCount -= 1;
		s->watchedDirty = true;
	}
}

int uploadWatched(s2amdSolver* s)
{
	if (!s->worldResident || s->structureDirty || s->contactCapacity <= 0 || (int)s->hContactWatched.size() != s->contactCapacity)
	{
		return S2AMD_OK; // (a rebuild writes the whole array)
	}
	// a structure that was built through s2amd_upload / s2amd_solve (worldResident false at the time) has watched slots on the
	// host only: the array is made when the world chain first needs it
	const bool missing = s->watchedCount > 0 && s->dWatched.bytes < (size_t)s->contactCapacity;
	if (!s->watchedDirty && !missing)
	{
		return S2AMD_OK;
	}
	if (s->dWatched.bytes < (size_t)s->contactCapacity)
	{
		bool grew = false;
		int rc = s->dWatched.ensure(std::max<size_t>((size_t)s->contactCapacity, 256), &grew);
		if (rc)
		{
			return rc;
		}
	}
	HIP_TRY(hipMemcpyAsync(s->dWatched.p, s->hContactWatched.data(), (size_t)s->contactCapacity, hipMemcpyHostToDevice, s->stream));
	HIP_TRY(hipStreamSynchronize(s->stream)); // the bytes are a std::vector the next call may touch
	s->watchedDirty = false;
	return S2AMD_OK;
}
